"""FaceID golden vectors (tests/golden/make_golden_faceid.py: outputs of the imported reference's ``FaceIDAttnProcessor``,
attn_processors.py:98-180; VERDICT r3 'missing' 4).  The branch is out of scope for every BASELINE config,
but the importable name runs its attention on the fused kernel (a ragged K/V tile of 1 / 4 / 16 face tokens, or the whole
token axis in self mode): pinned here.  CPU: the oracle restatement and the state_dict names against the fixture.  GPU: our
processor under autocast.  Tolerance (floating point): max(2 TOL max(1, |ref|), the reference's own 16-bit deviation on
the same inputs), TOL = 1e-3 (fp16) / 8e-3 (bf16) - the bound of tests/test_golden_r4.py."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import faceid_inputs as FI  # noqa: E402

from oracle import shared_attn_oracle as O  # noqa: E402

Z = np.load(os.path.join(HERE, "golden", "instantrestore_golden_faceid.npz"))
MANIFEST = json.loads(bytes(Z["manifest"]).decode())
TOL = {"f16": 1e-3, "bf16": 8e-3}


def _inputs(m):
    d = FI.build(m)
    assert abs(FI.checksum(d) - m["checksum"]) <= 1e-6 * abs(m["checksum"]), \
        "seeded inputs differ from the ones the reference was run on (torch RNG drift?): regenerate the fixture"
    return d


def test_fixture_covers_cross_and_self_mode():
    assert [m["id"] for m in MANIFEST] == [c["id"] for c in FI.CASES]
    assert sorted(m["is_self_attn"] for m in MANIFEST) == [False, False, False, True]


@pytest.mark.parametrize("m", MANIFEST, ids=[m["id"] for m in MANIFEST])
def test_oracle_matches_the_reference_faceid_processor(m):
    d = _inputs(m)
    f = lambda t: t.numpy().astype(np.float64)
    out = O.faceid_processor_np(f(d["hidden"]), f(d["wq"]), f(d["wo"]), f(d["bo"]), f(d["wp"]), f(d["bp"]), f(d["wk"]), f(d["wv"]),
                                m["H"], encoder_hidden=f(d["encoder"]) if "encoder" in d else None)
    ref = Z[f"{m['id']}/out"].astype(np.float64)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max())   # the reference ran in fp32


@pytest.mark.parametrize("m", MANIFEST, ids=[m["id"] for m in MANIFEST])
def test_state_dict_names_are_the_reference_ones(m):
    from face_replace.models.attn_processors import FaceIDAttnProcessor
    p = FaceIDAttnProcessor(m["C"], self_attn_idx=None, cross_attention_dim=m["cross"], embed_dim=m["embed"])
    assert sorted(p.state_dict().keys()) == m["state_dict_keys"]
    width = m["cross"] or m["C"]
    assert p.face_projection.weight.shape == (width, m["embed"]) and p.to_k_face_embed.weight.shape == (m["C"], width)


@pytest.mark.gpu
@pytest.mark.parametrize("m", MANIFEST, ids=[m["id"] for m in MANIFEST])
def test_faceid_through_our_processor(m):
    from face_replace.models.attn_processors import FaceIDAttnProcessor
    from instantrestore_amd.attention import Attention
    d = _inputs(m)
    dtype = FI.TORCH_DT[m["lowp"]]
    attn = Attention(query_dim=m["C"], cross_attention_dim=None, heads=m["H"], dim_head=64)
    proc = FaceIDAttnProcessor(m["C"], self_attn_idx=None, cross_attention_dim=m["cross"], embed_dim=m["embed"])
    with torch.no_grad():
        attn.to_q.weight.copy_(d["wq"]); attn.to_out[0].weight.copy_(d["wo"]); attn.to_out[0].bias.copy_(d["bo"])
        proc.face_projection.weight.copy_(d["wp"]); proc.face_projection.bias.copy_(d["bp"])
        proc.to_k_face_embed.weight.copy_(d["wk"]); proc.to_v_face_embed.weight.copy_(d["wv"])
    attn.set_processor(proc)
    attn = attn.cuda()
    enc = d["encoder"].cuda() if "encoder" in d else None
    with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
        # ref_keys / ref_values are handed to every processor by the pipeline and ignored here (attn_processors.py:123-124)
        out = attn(d["hidden"].cuda(), encoder_hidden_states=enc, ref_keys=[torch.zeros(1, 1, 8, m["C"], device="cuda", dtype=dtype)],
                   ref_values=[torch.zeros(1, 1, 8, m["C"], device="cuda", dtype=dtype)])
    assert out.shape == d["hidden"].shape and out.dtype == dtype
    assert proc.is_self_attn == m["is_self_attn"]
    ref = Z[f"{m['id']}/out"].astype(np.float64)
    ref_lowp_err = np.abs(Z[f"{m['id']}/out_lowp"].astype(np.float64) - ref).max()
    err = np.abs(out.float().cpu().numpy() - ref).max()
    bound = max(2 * TOL[m["lowp"]] * max(1.0, np.abs(ref).max()), ref_lowp_err)
    assert err <= bound, (m["id"], err, bound, ref_lowp_err)
