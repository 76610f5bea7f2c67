"""Attention mass per K/V segment as a BY-PRODUCT of the attention launch (ABI v9 ``ir_shared_attn_args.seg_mass``,
``ops.shared_attention(..., return_mass=True)``): what gradio_demo.py:119-127 reduces ``attention_probs`` (attn_processors.py:258-261)
to, without the tensor and without a second pass over Q and K.

The forward kernels store the cumulative log-sum-exp at every segment boundary of their K/V walk and a row-sized kernel turns the
differences into masses.  Checked here: against the float64 oracle's block sums of the probability matrix on identical
16-bit-rounded inputs (abs 2e-3, the bound of tests/test_gpu_probs.py::test_segment_mass: the inputs' rounding is the only 16-bit
quantity on the way), against the second-pass kernel ``ir_attn_segment_mass`` (abs 1e-4: same fp32 arithmetic, another summation
order), rows summing to 1, the attention output BIT-identical to the launch without the by-product, with and without the AdaIN
fold, the self segment, pre-scaled Q, ragged segment lengths, zero-filled references closed analytically (``valid_refs``) and the
K/V-range pieces of the remainder split (the cfg-2 shapes of both kernels)."""
import os

import numpy as np
import pytest
import torch

from oracle import shared_attn_oracle as O
from parity_bounds import check_parity

pytestmark = pytest.mark.gpu

QC = 0.125 * 1.4426950408889634


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from instantrestore_amd import ops as _ops
    _ops._lib.lib()
    return _ops


def _rand(shape, dtype, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


def _np64(t):
    return t.float().cpu().numpy().astype(np.float64)


def _edges(Ls, N, Lr, inc):
    return [0] + ([Ls] if inc else []) + [(Ls if inc else 0) + (n + 1) * Lr for n in range(N)]


def _block_sums(p, edges):
    return np.stack([p[..., a:b].sum(-1) for a, b in zip(edges[:-1], edges[1:])], axis=-1)


CASES = [
    # B, H, Lq, Ls, N, Lr, include_self
    (2, 2, 72, 72, 3, 40, True),        # ragged 64-key tiles
    (2, 2, 72, 72, 3, 40, False),
    (1, 3, 256, 256, 4, 256, True),     # the 16x16-token class, thin
    (1, 2, 300, 304, 2, 136, True),     # cross-length self segment, key tails
    (2, 2, 33, 33, 2, 37, True),        # nothing aligned
    (1, 1, 77, 77, 1, 1, True),         # a one-token reference
    (2, 1, 64, 64, 8, 64, True),        # eight references
    (1, 2, 96, 96, 0, 0, True),         # no references: one segment, mass 1
    (1, 1, 1024, 1024, 4, 1024, True),  # the 32x32-token class, one head
    (1, 1, 4096, 4096, 2, 192, True),   # 4096 query rows: the 64-row kernel (forced below where the default rule would not take it)
]


def _ids(cases):
    return [f"B{c[0]}H{c[1]}Lq{c[2]}Ls{c[3]}N{c[4]}Lr{c[5]}{'s' if c[6] else 'n'}" for c in cases]


def _inputs(case, dtype, seed):
    B, H, Lq, Ls, N, Lr, inc = case
    C = H * 64
    gen = torch.Generator().manual_seed(seed)
    q = _rand((B, Lq, C), dtype, gen, 1.5)
    k, v = _rand((B, Ls, C), dtype, gen, 1.5), _rand((B, Ls, C), dtype, gen)
    rk = _rand((B, N, Lr, C), dtype, gen, 1.5) if N else None
    rv = _rand((B, N, Lr, C), dtype, gen) if N else None
    return q, k, v, rk, rv


@pytest.mark.parametrize("presc", [False, True], ids=["plainq", "prescq"])
@pytest.mark.parametrize("adain", [False, True], ids=["noadain", "adain"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=_ids(CASES))
def test_by_product_mass_against_the_oracle(ops, case, dtype, adain, presc):
    B, H, Lq, Ls, N, Lr, inc = case
    if adain and (N == 0 or Lr < 8):
        pytest.skip("AdaIN needs references (and the fold's stated regime, DESIGN section 2)")
    q, k, v, rk, rv = _inputs(case, dtype, seed=21)
    scale = 0.125
    if presc:   # what the fused q/k/v projection hands over: Q * scale * log2(e), rounded once
        q = (q.float() * QC).to(dtype)
        oracle_scale = 0.6931471805599453
    else:
        oracle_scale = scale
    _, p_ref = O.shared_attention_np(_np64(q), _np64(k), _np64(v), None if rk is None else _np64(rk), None if rv is None else _np64(rv),
                                     H, oracle_scale, False, inc, return_probs=True)
    m_ref = _block_sums(p_ref.reshape(B, H, Lq, -1), _edges(Ls, N, Lr, inc))
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    rkd, rvd = (rk.cuda(), rv.cuda()) if N else (None, None)
    aff = ops.adain_stats(vd, rvd, heads=H) if adain else None
    kw = dict(heads=H, scale=scale, include_self=inc, adain=aff, q_prescaled=presc)
    forced = Lq >= 4096
    if forced:
        ops.set_attn_variant(13)   # IR_TUNE_W64X8: one head does not fill the chip, the default rule would take the 32-row kernel
    try:
        out_plain, lse = ops.shared_attention(qd, kd, vd, rkd, rvd, return_lse=True, **kw)
        out, lse2, mass = ops.shared_attention(qd, kd, vd, rkd, rvd, return_lse=True, return_mass=True, **kw)
        if forced:
            assert "w64" in ops.shared_attention_kernel_name(qd, kd, vd, rkd, rvd, heads=H, scale=scale, include_self=inc, adain=aff, q_prescaled=presc)
    finally:
        ops.set_attn_variant(0)
    assert torch.equal(out, out_plain) and torch.equal(lse, lse2), "the by-product changed the attention result"
    assert mass.dtype == torch.float32 and tuple(mass.shape) == m_ref.shape
    m = mass.cpu().numpy()
    assert np.isfinite(m).all()
    assert np.abs(m - m_ref).max() <= 2e-3, np.abs(m - m_ref).max()
    assert np.abs(m.sum(-1) - 1.0).max() <= 1e-5                       # the differences telescope
    assert m.min() >= -1e-6
    second = ops.attn_segment_mass(qd, kd, rkd, lse, heads=H, scale=scale, include_self=inc, q_prescaled=presc)
    assert float((mass - second).abs().max()) <= 1e-4, float((mass - second).abs().max())


@pytest.mark.parametrize("adain", [False, True], ids=["noadain", "adain"])
@pytest.mark.parametrize("inc", [False, True], ids=["noself", "self"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_mass_with_zero_filled_references_closed_analytically(ops, dtype, inc, adain):
    """``valid_refs``: the all-zero suffix of the reference list is not walked - its segments' masses come from the closed form
    (Lr keys of weight 2^(-m) each) and must equal the oracle's on the zero-filled tensors: zeroed, not masked, the zero segments
    soak up mass (~65 % in the survey's example)"""
    B, H, L, N, Lr = 3, 2, 200, 4, 72
    C = H * 64
    gen = torch.Generator().manual_seed(33)
    q = _rand((B, L, C), dtype, gen, 1.5)
    k, v = _rand((B, L, C), dtype, gen, 1.5), _rand((B, L, C), dtype, gen)
    rk, rv = _rand((B, N, Lr, C), dtype, gen, 1.5), _rand((B, N, Lr, C), dtype, gen)
    valid = [4, 1, 0]                                # 0 valid references without a self segment: every key of that entry is zero
    for b, nv in enumerate(valid):
        rk[b, nv:] = 0
        rv[b, nv:] = 0
    qd, kd, vd, rkd, rvd = (t.cuda() for t in (q, k, v, rk, rv))
    vt = torch.tensor(valid, dtype=torch.int32, device="cuda")
    aff = None
    rv_eff = _np64(rv)
    if adain:
        aff = ops.adain_stats(vd, rvd, heads=H)
        a, bsh = (t.cpu().numpy().astype(np.float64).reshape(B, N, 1, C) for t in aff)
        rv_eff = rv_eff * a + bsh
    out_ref, p_ref = O.shared_attention_np(_np64(q), _np64(k), _np64(v), _np64(rk), rv_eff, H, 0.125, False, inc, return_probs=True)
    m_ref = _block_sums(p_ref.reshape(B, H, L, -1), _edges(L, N, Lr, inc))
    kw = dict(heads=H, scale=0.125, include_self=inc, adain=aff)
    walked = ops.shared_attention(qd, kd, vd, rkd, rvd, return_mass=True, **kw)
    closed = ops.shared_attention(qd, kd, vd, rkd, rvd, return_mass=True, valid_refs=vt, **kw)
    for out, mass in (walked, closed):
        m = mass.cpu().numpy()
        assert np.abs(m - m_ref).max() <= 2e-3, np.abs(m - m_ref).max()
        assert np.abs(m.sum(-1) - 1.0).max() <= 1e-5
        check_parity(out, out_ref, dtype, "seg_mass launch output")
    assert float((walked[1] - closed[1]).abs().max()) <= 1e-5
    if not inc:
        # every key of batch entry 2 is zero: each of its N segments holds exactly 1/N of every row
        np.testing.assert_allclose(closed[1][2].cpu().numpy(), 1.0 / N, atol=1e-6)


SPLIT_SHAPES = [
    # the cfg-2 shapes whose last round of work items is cut into K/V-range pieces (workspace given): B, H, L, N
    ("32x32 tokens, 32-row kernel", 8, 10, 1024, 4),
    ("64x64 tokens, 64-row kernel", 8, 5, 4096, 4),
]


@pytest.mark.parametrize("presc", [True, False], ids=["prescq-bf16", "plainq-f16"])
@pytest.mark.parametrize("t", [0, 1], ids=["noself", "self"])
@pytest.mark.parametrize("adain", [False, True], ids=["noadain", "adain"])
@pytest.mark.parametrize("shape", SPLIT_SHAPES, ids=["L1024", "L4096"])
def test_mass_through_the_remainder_split_at_the_cfg2_shapes(ops, shape, adain, t, presc):
    """At cfg 2 both kernels cut the items of their last, partially filled round into K/V-range pieces whose partial results a
    second kernel merges; a piece knows the cumulative sums of ITS key range only, the merge adds them up.  Compared with the
    second-pass kernel (itself oracle-checked, tests/test_gpu_probs.py) at the real shape, pre-scaled Q in bf16 as the processors launch
    it and plain Q in fp16, with every count of valid references in the batch."""
    _, B, H, L, N = shape
    C = H * 64
    dtype = torch.bfloat16 if presc else torch.float16
    torch.manual_seed(5)
    q = (torch.randn(B, L, C, device="cuda") * 1.2 * (QC if presc else 1.0)).to(dtype)
    k, v = (torch.randn(B, L, C, device="cuda") * 1.2).to(dtype), torch.randn(B, L, C, device="cuda").to(dtype)
    rk, rv = (torch.randn(B, N, L, C, device="cuda") * 1.2).to(dtype), torch.randn(B, N, L, C, device="cuda").to(dtype)
    aff = ops.adain_stats(v, rv, heads=H) if adain else None
    kw = dict(heads=H, scale=0.125, include_self=bool(t), adain=aff, q_prescaled=presc)
    out0, lse = ops.shared_attention(q, k, v, rk, rv, return_lse=True, **kw)
    out, mass = ops.shared_attention(q, k, v, rk, rv, return_mass=True, **kw)
    nosplit = ops.shared_attention(q, k, v, rk, rv, return_mass=True, split=False, **kw)[1]
    second = ops.attn_segment_mass(q, k, rk, lse, heads=H, scale=0.125, include_self=bool(t), q_prescaled=presc)
    # the output of a launch that also leaves the masses is the output of the same kernel without them, bit for bit.  Round 6:
    # where the default dispatch takes the 128-row kernel (pre-scaled Q, L >= 4096) the masses come from the 64-row kernel
    # (the 128-row kernel has no such form), so "the same kernel" is tuning 13 there and the default output agrees to rounding
    if "w128" in ops.shared_attention_kernel_name(q, k, v, rk, rv, **kw):
        prev = ops.set_attn_variant(13)
        try:
            out13 = ops.shared_attention(q, k, v, rk, rv, **kw)
        finally:
            ops.set_attn_variant(prev)
        assert torch.equal(out, out13)
        assert float((out.float() - out0.float()).abs().max()) <= 2 * 2.0 ** -8 * float(out0.float().abs().max()) + 4e-4
    else:
        assert torch.equal(out, out0)
    assert float((mass - second).abs().max()) <= 1e-4, float((mass - second).abs().max())
    assert float((nosplit - second).abs().max()) <= 1e-4
    assert float((mass.sum(-1) - 1).abs().max()) <= 1e-5
    # ragged valid counts: zero the suffixes, pass the counts, compare with the walk over the zero tiles
    valid = torch.tensor([(b % (N + 1)) for b in range(B)], dtype=torch.int32, device="cuda")
    # (entry 0 and entry N + 1 have NO valid reference: without the self segment every key of theirs is zero, the item owns no tile at all)
    for b in range(B):
        rk[b, int(valid[b]):] = 0
        rv[b, int(valid[b]):] = 0
    aff = ops.adain_stats(v, rv, heads=H) if adain else None
    kw["adain"] = aff
    walked = ops.shared_attention(q, k, v, rk, rv, return_mass=True, **kw)[1]
    closed = ops.shared_attention(q, k, v, rk, rv, return_mass=True, valid_refs=valid, **kw)[1]
    # (the walk adds up to 16 384 equal terms per zero segment one by one in fp32 - the closed form is the exact product: the plain-Q
    #  kernels, which add every probability straight into the running sums, come out up to 4e-5 apart, the pre-scaled ones 1e-5)
    assert float((walked - closed).abs().max()) <= 1e-4, float((walked - closed).abs().max())
    assert float((closed.sum(-1) - 1).abs().max()) <= 1e-5


def test_kernels_without_the_by_product_refuse_it(ops):
    """only the kernels of the default dispatch carry the instantiation: an explicitly tuned other kernel says so instead of
    returning an unwritten buffer"""
    q = torch.randn(1, 128, 64, device="cuda").to(torch.bfloat16)
    rk = torch.randn(1, 2, 128, 64, device="cuda").to(torch.bfloat16)
    ops.set_attn_variant(10)
    try:
        with pytest.raises(RuntimeError):
            ops.shared_attention(q, q, q, rk, rk, heads=1, scale=0.125, return_mass=True)
        ops.shared_attention(q, q, q, rk, rk, heads=1, scale=0.125)
    finally:
        ops.set_attn_variant(0)


def test_seeded_sweep_of_by_product_masses(ops):
    """random small shapes, both kernels' default forms, every flag combination: by-product vs second pass
    (IR_SWEEP_CASES / IR_SWEEP_SEED / IR_SWEEP_MAXLQ / IR_SWEEP_MAXLR widen it for a soak)"""
    seed = int(os.environ.get("IR_SWEEP_SEED", "2029"))
    max_lq, max_lr = int(os.environ.get("IR_SWEEP_MAXLQ", "400")), int(os.environ.get("IR_SWEEP_MAXLR", "300"))
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    for case in range(int(os.environ.get("IR_SWEEP_CASES", "60"))):
        B, H = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        Lq = int(rng.integers(1, max_lq))
        N = int(rng.integers(0, 6))
        Lr = int(rng.integers(8, max_lr)) if N else 0
        inc = bool(rng.integers(0, 2)) or N == 0
        Ls = Lq if rng.integers(0, 2) else int(rng.integers(1, max_lr))
        adain = bool(N and rng.integers(0, 2))
        presc = bool(rng.integers(0, 2))
        dtype = [torch.float16, torch.bfloat16][case % 2]
        C = H * 64
        q = _rand((B, Lq, C), dtype, gen, 1.4).cuda()
        if presc:
            q = (q.float() * QC).to(dtype)
        k, v = _rand((B, Ls, C), dtype, gen, 1.4).cuda(), _rand((B, Ls, C), dtype, gen).cuda()
        rk = _rand((B, N, Lr, C), dtype, gen, 1.4).cuda() if N else None
        rv = _rand((B, N, Lr, C), dtype, gen).cuda() if N else None
        vt = None
        if N and rng.integers(0, 2):
            vt = torch.tensor(rng.integers(0 if inc else 1, N + 1, size=B), dtype=torch.int32, device="cuda")
            for b in range(B):
                rk[b, int(vt[b]):] = 0
                rv[b, int(vt[b]):] = 0
        what = f"case {case}: B{B} H{H} Lq{Lq} Ls{Ls} N{N} Lr{Lr} inc{inc} adain{adain} presc{presc} valid{None if vt is None else vt.tolist()} {dtype}"
        aff = ops.adain_stats(v, rv, heads=H) if adain else None
        kw = dict(heads=H, scale=0.125, include_self=inc, adain=aff, q_prescaled=presc)
        out0, lse = ops.shared_attention(q, k, v, rk, rv, return_lse=True, valid_refs=vt, **kw)
        out, mass = ops.shared_attention(q, k, v, rk, rv, return_mass=True, valid_refs=vt, **kw)
        assert torch.equal(out, out0), what
        second = ops.attn_segment_mass(q, k, rk, lse, heads=H, scale=0.125, include_self=inc, q_prescaled=presc)
        assert float((mass - second).abs().max()) <= 1e-4, (what, float((mass - second).abs().max()))
        assert float((mass.sum(-1) - 1).abs().max()) <= 1e-5, what
