"""Zero-filled references in closed form (ABI v8 ``valid_refs``).

``Pix2Pix_Turbo.get_conditioning_keys_values`` zero-fills the references ``n >= valid_indices[b]`` in place
(pix2pix_turbo.py:269-273): zeroed, NOT masked - their keys score exactly 0 and keep an exp(0) softmax weight, their value
rows are 0 (with AdaIN: the style mean).  When the caller passes the valid counts, the default kernels do not walk that suffix
of the reference list: row sum and output take its contribution analytically.  Held to: the float64 oracle evaluated on the
ZERO-FILLED tensors (the reference's own semantics) within the tolerance of tests/test_gpu_parity.py, and to the same call
without ``valid_refs`` (which walks the zero tiles) within twice that tolerance.  SURVEY 8d: ``inference/test.py:81`` passes
``max_conditioning_images`` of the CHECKPOINT as valid count - 4 of 8 in the cfg-4 stress case."""
import numpy as np
import pytest
import torch

from oracle import shared_attn_oracle as O
from parity_bounds import check_parity

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from instantrestore_amd import ops as _ops
    _ops._lib.lib()
    return _ops


def _np64(t):
    return t.float().cpu().numpy().astype(np.float64)


def _case(B, H, L, N, Lr, dtype, seed, scale_q=1.0):
    g = torch.Generator().manual_seed(seed)
    C = H * 64
    q = (torch.randn(B, L, C, generator=g) * scale_q).to(dtype)
    k, v = torch.randn(B, L, C, generator=g).to(dtype), (torch.randn(B, L, C, generator=g) * 0.8 + 0.3).to(dtype)
    rk = torch.randn(B, N, Lr, C, generator=g).to(dtype)
    rv = (torch.randn(B, N, Lr, C, generator=g) * 1.3 - 0.2).to(dtype)
    return q, k, v, rk, rv


def _run(ops, q, k, v, rk, rv, valid, H, inc, adain, variant=0, pass_valid=True):
    rk, rv = rk.clone().cuda(), rv.clone().cuda()
    vd = torch.tensor(valid, dtype=torch.int32, device="cuda")
    ops.zero_invalid_refs(rk, rv, vd, heads=H)
    qd, kd, vvd = q.cuda(), k.cuda(), v.cuda()
    aff = ops.adain_stats(vvd, rv, heads=H) if adain else None
    ops.set_attn_variant(variant)
    try:
        out, lse = ops.shared_attention(qd, kd, vvd, rk, rv, heads=H, scale=0.125, include_self=inc, adain=aff, return_lse=True,
                                        valid_refs=vd if pass_valid else None)
    finally:
        ops.set_attn_variant(0)
    return out, lse, rk, rv


SMALL = [
    # B, H, L, N, Lr, include_self, valid
    (3, 2, 256, 4, 256, True, [4, 2, 0]),        # 16x16-token class: all valid / half / none (self segment only)
    (3, 2, 256, 4, 256, False, [1, 3, 0]),       # no self segment: valid 0 = ONLY zero keys (uniform weights over nothing but zeros)
    (2, 1, 1024, 4, 1024, True, [3, 1]),         # 32x32-token class
    (2, 3, 200, 5, 72, True, [2, 5]),            # ragged tiles, partial query block
    (9, 2, 1024, 4, 1024, True, [4, 3, 2, 1, 0, 1, 2, 3, 4]),   # remainder split of the item grid with per-item K/V ranges
    # a short self segment before long references, none of them valid: the split plans its pieces from the FULL tile count (2 + 5 * 6),
    # the item has two tiles - some pieces are EMPTY, and one of them "stops inside" the self segment.  (Found by the widened
    # sweep of tests/test_gpu_seg_mass.py: the 32-row kernel's fold of such a piece took 2^(-inf - -inf) and returned NaN rows.)
    (1, 2, 97, 5, 373, True, [0]),
    (2, 2, 128, 5, 373, True, [0, 1]),
    (2, 1, 70, 5, 373, False, [0, 1]),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("adain", [False, True], ids=["plain", "adain"])
@pytest.mark.parametrize("case", SMALL, ids=[f"B{c[0]}H{c[1]}L{c[2]}N{c[3]}Lr{c[4]}s{int(c[5])}" for c in SMALL])
def test_closed_form_matches_the_oracle_on_zero_filled_references(ops, case, adain, dtype):
    B, H, L, N, Lr, inc, valid = case
    q, k, v, rk, rv = _case(B, H, L, N, Lr, dtype, seed=31 + L + N)
    out, lse, rkz, rvz = _run(ops, q, k, v, rk, rv, valid, H, inc, adain)
    for b in range(B):
        assert float(rkz[b, valid[b]:].abs().max() if valid[b] < N else 0.0) == 0.0
    ref, p_ref = O.shared_attention_np(_np64(q), _np64(k), _np64(v), _np64(rkz), _np64(rvz), H, 0.125, adain, inc, return_probs=True)
    got = _np64(out)
    assert np.isfinite(got).all()
    bound = TOL[dtype] * max(1.0, np.abs(ref).max())
    check_parity(got, ref, dtype, "valid_refs closed form vs the oracle on the zero-filled tensors")
    walked, lse_w, _, _ = _run(ops, q, k, v, rk, rv, valid, H, inc, adain, pass_valid=False)
    assert np.abs(got - _np64(walked)).max() <= 2 * bound
    # the LSE carries the zero keys too: logsumexp over ALL columns of the zero-filled problem
    assert float((lse - lse_w).abs().max()) <= 2e-3


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("adain", [False, True], ids=["plain", "adain"])
def test_cfg4_top_layer_four_of_eight_references_valid(ops, adain, dtype):
    """the cfg-4 stress shape (8 references, 64x64 tokens, Lkv = 36 864) with the checkpoint's 4 as valid count: the 64-row kernel;
    sampled rows against the oracle"""
    B, H, L, N = 2, 2, 4096, 8
    q, k, v, rk, rv = _case(B, H, L, N, L, dtype, seed=77)
    valid = [4, 6]
    out, lse, rkz, rvz = _run(ops, q, k, v, rk, rv, valid, H, True, adain)
    rows = np.array(sorted(set(np.random.default_rng(3).integers(0, L, 96).tolist() + [0, 63, 64, L - 1])))
    ref = O.shared_attention_np(_np64(q)[:, rows], _np64(k), _np64(v), _np64(rkz), _np64(rvz), H, 0.125, adain, True)
    got = _np64(out)[:, rows]
    bound = TOL[dtype] * max(1.0, np.abs(ref).max())
    check_parity(got, ref, dtype, "valid_refs closed form vs the oracle on the zero-filled tensors")
    walked, _, _, _ = _run(ops, q, k, v, rk, rv, valid, H, True, adain, pass_valid=False)
    assert np.abs(_np64(out) - _np64(walked)).max() <= 2 * bound


def test_reference_far_below_zero_moves_up_to_the_zero_score(ops):
    """every real score is hugely negative (q ~ -k): the zero keys own the softmax; the closed form must move the running
    reference up to 0 instead of weighing the zero keys with 2^(+large)"""
    dtype = torch.bfloat16
    B, H, L, N = 1, 1, 256, 2
    g = torch.Generator().manual_seed(5)
    base = torch.randn(B, L, 64, generator=g)
    q, k = (base * 3).to(dtype), (-base * 3).to(dtype)
    v = torch.randn(B, L, 64, generator=g).to(dtype)
    rk = (-base * 3).reshape(B, 1, L, 64).repeat(1, N, 1, 1).to(dtype)
    rv = torch.randn(B, N, L, 64, generator=g).to(dtype)
    for adain in (False, True):
        out, lse, rkz, rvz = _run(ops, q, k, v, rk, rv, [1], H, True, adain)
        ref = O.shared_attention_np(_np64(q), _np64(k), _np64(v), _np64(rkz), _np64(rvz), H, 0.125, adain, True)
        got = _np64(out)
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() <= TOL[dtype] * max(1.0, np.abs(ref).max())


def test_processor_takes_the_valid_counts_from_the_harvest(ops):
    """kv_harvest(with_valid=True) -> cross_attention_kwargs['ref_valid'] -> SharedAttnProcessor: same tokens as the walk"""
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    torch.manual_seed(9)
    B, H, L, N = 4, 2, 256, 4
    C = H * 64
    attn = Attention(query_dim=C, heads=H, dim_head=64,
                     processor=SharedAttnProcessor(self_attn_idx=0, use_adain=True, train_input=True)).cuda().to(torch.bfloat16)
    x = torch.randn(B, L, C, device="cuda", dtype=torch.bfloat16)
    rk = torch.randn(B, N, L, C, device="cuda", dtype=torch.bfloat16)
    rv = torch.randn(B, N, L, C, device="cuda", dtype=torch.bfloat16)
    valid = torch.tensor([4, 2, 1, 3], dtype=torch.int32, device="cuda")
    ops.zero_invalid_refs(rk, rv, valid, heads=H)
    with torch.no_grad():
        a = attn(x, ref_keys=[rk], ref_values=[rv], ref_valid=valid)
        b = attn(x, ref_keys=[rk], ref_values=[rv])
    assert float((a.float() - b.float()).abs().max()) <= 2 * 8e-3 * max(1.0, float(b.float().abs().max()))


def test_randomised_valid_counts_equal_the_walk(ops):
    """seeded sweep (IR_SWEEP_CASES widens it): ragged shapes, random valid counts per batch entry, with and without the self
    segment and AdaIN: telling the kernel the counts must give what walking the zero-filled references gives (2 x TOL: another
    fp32 summation order), on every product kernel that implements the closed form and on those that ignore the promise"""
    import os
    rng = np.random.default_rng(int(os.environ.get("IR_SWEEP_SEED", "91")))
    for case in range(int(os.environ.get("IR_SWEEP_CASES", "24"))):
        B, H = int(rng.integers(1, 5)), int(rng.integers(1, 3))
        L = int(rng.integers(8, 600))
        N = int(rng.integers(1, 7))
        Lr = int(rng.integers(8, 300))
        if case % 3 == 2:   # short query / self axis before long references: items that own far fewer tiles than the split planned for
            L, Lr = int(rng.integers(8, 260)), int(rng.integers(200, 700))
        inc = bool(rng.integers(0, 2))
        adain = bool(rng.integers(0, 2))
        dtype = [torch.float16, torch.bfloat16][case % 2]
        valid = [int(x) for x in rng.integers(0, N + 1, B)]
        if case % 6 == 5:
            valid = [int(x) for x in rng.integers(0, 2, B)]   # (almost) nothing valid
        variant = [0, 0, 10, 13, 7][case % 5]          # default dispatch, 32-row kernels, 64-row kernel, exact-max form (walks)
        q, k, v, rk, rv = _case(B, H, L, N, Lr, dtype, seed=500 + case)
        a, lse_a, _, _ = _run(ops, q, k, v, rk, rv, valid, H, inc, adain, variant=variant)
        b, lse_b, _, _ = _run(ops, q, k, v, rk, rv, valid, H, inc, adain, variant=variant, pass_valid=False)
        what = f"case {case}: B{B} H{H} L{L} N{N} Lr{Lr} inc{inc} adain{adain} valid{valid} variant{variant} {dtype}"
        bound = 2 * TOL[dtype] * max(1.0, float(b.float().abs().max()))
        assert torch.isfinite(a.float()).all(), what
        assert float((a.float() - b.float()).abs().max()) <= bound, what
        assert float((lse_a - lse_b).abs().max()) <= 2e-3, what


@pytest.mark.parametrize("fused", [True, False], ids=["gemm_partials", "standalone_statistics"])
def test_two_unet_host_with_zero_filled_references_statistics_and_counts(fused):
    """the whole plugin surface on the topology host: reference UNet -> harvest(with_stats, with_valid) with valid = [2, 1] of 3
    (zero fill in place, statistics of the zeroed references invalidated) -> main UNet with 'ref_stats' and 'ref_valid' against the
    same call without 'ref_valid' (the kernels walk the zero tiles): same latents within twice the bf16 tolerance.  Both forms of
    the content statistics: the q/k/v GEMM's partials (round 4) and the standalone pass."""
    from types import SimpleNamespace
    from face_replace.models.attn_processors import register_attention_processor, register_attention_processor_kv_unet
    from instantrestore_amd import attn_processors as ap
    from instantrestore_amd.kv_harvest import get_conditioning_keys_values
    from instantrestore_amd.unet_host import AttnTopologyUNet
    import __graft_entry__ as ge
    cfg = SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False)
    kv_unet, unet = AttnTopologyUNet(seed=5).to("cuda"), AttnTopologyUNet(seed=6).to("cuda")
    ge.register_attention_processor_kv_unet_default(kv_unet, cfg)
    register_attention_processor_kv_unet(kv_unet)
    register_attention_processor(unet, cfg)
    B, N, S = 2, 3, 32                                            # 32 x 32 latent: token axes of 16 / 64 / 256 ... per class
    text = torch.randn(1, 77, 1024, device="cuda")
    x = torch.randn(B, 4, S, S, device="cuda")
    ap.FUSED_STATS = fused
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            keys, vals, stats, valid = get_conditioning_keys_values(kv_unet, torch.randn(B * N, 4, S, S, device="cuda"), None,
                                                                    text.repeat(B * N, 1, 1), N, [2, 1], with_stats=True, with_valid=True)
            assert valid.tolist() == [2, 1] and all(float(k[1, 1:].abs().max()) == 0.0 for k in keys)
            kw = {"ref_keys": keys, "ref_values": vals, "ref_stats": stats}
            a = unet(x, None, encoder_hidden_states=text.repeat(B, 1, 1), cross_attention_kwargs=dict(kw, ref_valid=valid)).sample
            b = unet(x, None, encoder_hidden_states=text.repeat(B, 1, 1), cross_attention_kwargs=kw).sample
    finally:
        ap.FUSED_STATS = True
    torch.cuda.synchronize()
    assert torch.isfinite(a).all()
    assert float((a.float() - b.float()).abs().max()) <= 2 * 8e-3 * max(1.0, float(b.float().abs().max()))
