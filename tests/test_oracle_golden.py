"""Pin the oracle: the CPU restatement (oracle/) must reproduce the outputs the reference
itself produced (tests/golden/instantrestore_golden.npz, made by tests/golden/make_golden.py
from /root/reference/face_replace/models/attn_processors.py)."""
import numpy as np
import pytest

from conftest import GOLDEN_MANIFEST
from oracle import shared_attn_oracle as O

SHARED = [m for m in GOLDEN_MANIFEST if m["kind"] == "shared"]
KVCAP = [m for m in GOLDEN_MANIFEST if m["kind"] == "kv_capture"]
ADAIN = [m for m in GOLDEN_MANIFEST if m["kind"] == "adain"]


def _shared_inputs(golden, m):
    g = lambda n: golden.arr(m, n)
    return dict(hidden=g("hidden"), wq=g("wq"), wk=g("wk"), wv=g("wv"), wo=g("wo"), bo=g("bo"),
                ref_k=g("ref_k"), ref_v=g("ref_v"), heads=m["H"], use_adain=m["use_adain"],
                train_input=m["train_input"], encoder_hidden=g("enc"))


@pytest.mark.parametrize("m", SHARED, ids=[m["id"] for m in SHARED])
def test_shared_processor_matches_reference(golden, m):
    kw = _shared_inputs(golden, m)
    out, probs, _ = O.shared_attn_processor_np(**kw, dtype=np.float64, return_probs=True)
    ref = golden.arr(m, "out")
    # reference ran in float32; float64 oracle agrees to float32 round-off
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(out - ref).max() <= 2e-5 * scale
    ref_p = golden.arr(m, "probs")
    if ref_p is not None:
        assert probs.shape == ref_p.shape
        assert np.abs(probs - ref_p).max() <= (3e-5 if m["peaky"] else 2e-6)  # fp32 reference round-off on large logits
        # column order: [self (iff train_input)] ++ ref0 ++ ...
        assert probs.shape[-1] == (m["N"] + int(m["train_input"])) * m["Lr"] if m["N"] else True
        np.testing.assert_allclose(probs.sum(-1), 1.0, atol=1e-9)


@pytest.mark.parametrize("m", SHARED, ids=[m["id"] for m in SHARED])
def test_float32_port_matches_reference(golden, m):
    """the torch-CPU port that bench.py times as cpu_baseline is the same function"""
    import torch

    kw = _shared_inputs(golden, m)
    if kw["encoder_hidden"] is not None:
        pytest.skip("port covers the self-attention path only")
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
    out = O.shared_attn_processor_port(t(kw["hidden"]), t(kw["wq"]), t(kw["wk"]), t(kw["wv"]), t(kw["wo"]),
                                       t(kw["bo"]), t(kw["ref_k"]), t(kw["ref_v"]), m["H"],
                                       m["use_adain"], m["train_input"]).numpy()
    ref = golden.arr(m, "out")
    assert np.abs(out - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("m", KVCAP, ids=[m["id"] for m in KVCAP])
def test_kv_capture_matches_reference(golden, m):
    g = lambda n: golden.arr(m, n)
    hidden = g("hidden")
    out, _, (q, k, v) = O.shared_attn_processor_np(hidden, g("wq"), g("wk"), g("wv"), g("wo"), g("bo"),
                                                   None, None, m["H"], return_probs=True)
    assert np.abs(out - g("out")).max() <= 2e-5 * max(1.0, np.abs(g("out")).max())
    # the stash is the PRE-head-split projection (attn_processors.py:73-74)
    assert k.shape == g("keys").shape == (m["BN"], m["L"], m["H"] * 64)
    assert np.abs(k - g("keys")).max() <= 1e-5 and np.abs(v - g("values")).max() <= 1e-5


@pytest.mark.parametrize("m", ADAIN, ids=[m["id"] for m in ADAIN])
def test_adain_matches_reference(golden, m):
    content, style = golden.arr(m, "content").astype(np.float64), golden.arr(m, "style").astype(np.float64)
    s_mean, s_std = O.token_stats_np(style)
    out = O.adain_np(content, s_mean, s_std + O.ADAIN_EPS)
    ref = golden.arr(m, "out")
    assert np.abs(out - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())
    if m["zero_content"]:
        # an all-zero reference V maps EXACTLY onto the style mean
        np.testing.assert_allclose(out[0], np.broadcast_to(s_mean[0], out[0].shape), rtol=0, atol=1e-12)


def test_adain_affine_form_equals_adain():
    """the (a, b) affine the HIP stats kernel emits is the same function as adain()"""
    rng = np.random.default_rng(3)
    B, N, L, H = 2, 3, 37, 2
    v_self = rng.standard_normal((B, L, H * 64)) * 0.8 + 0.3
    ref_v = rng.standard_normal((B, N, L, H * 64)) * 1.7 - 0.5
    ref_v[1, 2] = 0.0               # zero-filled reference (pix2pix_turbo.py:269-273)
    ref_v[0, 1, :, 5:9] = 0.4375    # constant, non-zero channels: content std exactly 0, mean not
    a, b = O.adain_affine_np(v_self, ref_v, H)
    s_mean, s_std = O.token_stats_np(v_self)
    for n in range(N):
        want = O.adain_np(ref_v[:, n], s_mean, s_std + O.ADAIN_EPS)
        got = ref_v[:, n] * a[:, n][:, None, :] + b[:, n][:, None, :]
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)
    # where the content std is exactly 0 the affine is (0, style mean): adain() returns the style mean there for any ratio
    assert np.all(a[1, 2] == 0) and np.array_equal(b[1, 2], s_mean[1, 0])
    assert np.all(a[0, 1, 5:9] == 0) and np.array_equal(b[0, 1, 5:9], s_mean[0, 0, 5:9])


def test_zero_fill_is_not_masking():
    rng = np.random.default_rng(5)
    B, N, L, H = 2, 3, 16, 1
    q, k, v = (rng.standard_normal((B, L, 64)) for _ in range(3))
    rk, rv = rng.standard_normal((B, N, L, 64)), rng.standard_normal((B, N, L, 64))
    rk0, rv0 = O.zero_fill_invalid_np(rk, [3, 1]), O.zero_fill_invalid_np(rv, [3, 1])
    assert np.all(rk0[1, 1:] == 0) and np.all(rk0[0] == rk[0])
    _, p = O.shared_attention_np(q, k, v, rk0, rv0, H, 0.125, False, True, return_probs=True)
    # zeroed keys get exp(0) weight: identical, non-zero columns within a row
    blk = p[1, 0, :, 2 * L:]
    assert np.all(blk > 0) and np.allclose(blk, blk[:, :1])
