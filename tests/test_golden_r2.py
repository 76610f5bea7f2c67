"""Round-2 golden vectors (tests/golden/make_golden_r2.py, outputs of the imported reference): real-shape slices of every
layer class of BASELINE.json's configs and the reference's own registration map.  CPU part: the oracle against them and
this build's registration functions against the map.  GPU part: the same slices through our processors."""
import json
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import slice_inputs as SI  # noqa: E402

from oracle import shared_attn_oracle as O  # noqa: E402

Z = np.load(os.path.join(HERE, "golden", "instantrestore_golden_r2.npz"))
MANIFEST = json.loads(bytes(Z["manifest"]).decode())
REGISTRATION = json.loads(bytes(Z["registration"]).decode())
TOL = {"f16": 1e-3, "bf16": 8e-3}


def _inputs(m):
    d = SI.build(m)
    assert abs(SI.checksum(d) - m["checksum"]) <= 1e-6 * abs(m["checksum"]), \
        "seeded inputs differ from the ones the reference was run on (torch RNG drift?): regenerate the fixture"
    return d


@pytest.mark.parametrize("m", MANIFEST, ids=[m["id"] for m in MANIFEST])
def test_oracle_matches_reference_on_real_shape_slices(m):
    if m["L"] > 4096 and m["N"] * m["L"] > 40000:
        pass  # 65664 keys x 128 rows in float64: still < 1 s
    d = _inputs(m)
    f = lambda t: t.numpy().astype(np.float64)
    out, probs, _ = O.shared_attn_processor_np(f(d["hidden"]), f(d["wq"]), f(d["wk"]), f(d["wv"]), f(d["wo"]), f(d["bo"]),
                                               f(d["ref_k"]), f(d["ref_v"]), SI.HEADS, use_adain=m["use_adain"],
                                               train_input=m["train_input"], dtype=np.float64, return_probs=True)
    ref = Z[f"{m['id']}/out"]
    assert np.abs(out - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max())
    p = probs[0, 0]
    assert np.abs(p[:, d["cols"].numpy()] - Z[f"{m['id']}/probs_cols"]).max() <= 2e-6
    t, N, L = int(m["train_input"]), m["N"], m["L"]
    edges = [0] + ([SI.ROWS] if t else []) + [t * SI.ROWS + (n + 1) * L for n in range(N)]
    mass = np.stack([p[:, a:b].sum(-1) for a, b in zip(edges[:-1], edges[1:])], axis=1)
    assert np.abs(mass - Z[f"{m['id']}/block_mass"]).max() <= 2e-5


def _our_map(unet):
    return [[n, type(p).__name__, p.self_attn_idx, getattr(p, "use_adain", None), getattr(p, "train_input", None),
             getattr(p, "save_self_attentions", None)] for n, p in unet.attn_processors.items()]


@pytest.mark.parametrize("tag,kw", [("base", dict(use_adain=False, train_input=False)),
                                    ("adain", dict(use_adain=True, train_input=True)),
                                    ("faceid", dict(use_adain=True, train_input=False, condition_on_face_embeds=True))])
def test_registration_equals_the_references_own_map(tag, kw):
    """name -> (class, self_attn_idx, flags) as produced by the REFERENCE's register_attention_processor on the same host
    (attn_processors.py:282-321), including the order the nine indices are handed out in"""
    from types import SimpleNamespace
    from face_replace.models.attn_processors import register_attention_processor
    from instantrestore_amd.unet_host import AttnTopologyUNet
    cfg = SimpleNamespace(**dict(dict(use_adain=False, train_input=True, condition_on_face_embeds=False), **kw))
    unet = AttnTopologyUNet(seed=0)
    register_attention_processor(unet, cfg, save_self_attentions=(tag == "adain"))
    assert _our_map(unet) == REGISTRATION["main_" + tag]
    idx = [r[2] for r in REGISTRATION["main_" + tag] if r[2] is not None]
    assert idx == list(range(9))


def test_kv_unet_registration_equals_the_references_own_map():
    from face_replace.models.attn_processors import SharedAttnProcessor, register_attention_processor_kv_unet
    from instantrestore_amd.unet_host import AttnTopologyUNet
    unet = AttnTopologyUNet(seed=0)
    default = SharedAttnProcessor(self_attn_idx=None)
    unet.set_attn_processor({n: default for n in unet.attn_processors})
    register_attention_processor_kv_unet(unet)
    assert [[n, type(p).__name__] for n, p in unet.attn_processors.items()] == REGISTRATION["kv_unet"]
    assert sum(1 for r in REGISTRATION["kv_unet"] if r[1] == "AttnProcessor") == 9


@pytest.mark.gpu
@pytest.mark.parametrize("m", MANIFEST, ids=[m["id"] for m in MANIFEST])
def test_real_shape_slices_through_our_processor(m):
    """128 query rows x the full K/V of the layer class (up to 65664 keys) through SharedAttnProcessor under autocast,
    against the reference's fp32 output; attention_probs (dump path) against the reference's block masses (K/V block
    order: [self] ++ ref 0 ++ ... ) and its probabilities at 64 seeded columns per row (order inside the blocks)."""
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    d = _inputs(m)
    dtype = SI.TORCH_DT[m["lowp"]]
    attn = Attention(query_dim=SI.C, heads=SI.HEADS, dim_head=64)
    with torch.no_grad():
        attn.to_q.weight.copy_(d["wq"]); attn.to_k.weight.copy_(d["wk"]); attn.to_v.weight.copy_(d["wv"])
        attn.to_out[0].weight.copy_(d["wo"]); attn.to_out[0].bias.copy_(d["bo"])
    attn = attn.cuda()
    proc = SharedAttnProcessor(self_attn_idx=0, save_self_attentions=True, use_adain=m["use_adain"], train_input=m["train_input"])
    attn.set_processor(proc)
    with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
        out = attn(d["hidden"].cuda(), ref_keys=[d["ref_k"].to(dtype).cuda()], ref_values=[d["ref_v"].to(dtype).cuda()])
    ref = Z[f"{m['id']}/out"].astype(np.float64)
    err = np.abs(out.float().cpu().numpy() - ref).max()
    ref_err = np.abs(Z[f"{m['id']}/out_lowp"] - ref).max()    # the reference's own 16-bit run on the same inputs
    assert err <= max(2 * TOL[m["lowp"]] * max(1.0, np.abs(ref).max()), ref_err), (err, ref_err)
    p = proc.attention_probs[0, 0].float().cpu().numpy()
    t, N, L = int(m["train_input"]), m["N"], m["L"]
    assert p.shape == (SI.ROWS, t * SI.ROWS + N * L)
    edges = [0] + ([SI.ROWS] if t else []) + [t * SI.ROWS + (n + 1) * L for n in range(N)]
    mass = np.stack([p[:, a:b].sum(-1) for a, b in zip(edges[:-1], edges[1:])], axis=1)
    assert np.abs(mass - Z[f"{m['id']}/block_mass"]).max() <= 4 * TOL[m["lowp"]]       # per-block attention mass
    assert np.abs(p[:, d["cols"].numpy()] - Z[f"{m['id']}/probs_cols"]).max() <= 4 * TOL[m["lowp"]] * max(Z[f"{m['id']}/probs_cols"].max(), 1e-2) + 1e-5
