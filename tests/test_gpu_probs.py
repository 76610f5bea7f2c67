"""The dump path on the GPU (attn_processors.py:258-261): every kernel of ``ir_attn_probs_ex`` and the opt-in per-segment mass
(``ir_attn_segment_mass``) against the float64 oracle's probability matrix on identical 16-bit-rounded inputs.

Tolerance (floating point): probabilities are in [0, 1], so the bound of tests/test_gpu_parity.py reads
``max|P - P_ref| <= TOL[dtype]`` (1e-3 fp16, 8e-3 bf16: one ulp of the 16-bit output at 1.0); rows sum to 1 within 4x that.
The kernels of one call are also compared with each other BIT FOR BIT: they evaluate the same expression
(exp2(fma(s, scale*log2e, -lse*log2e)) on the fp32 MFMA result), only the store shape differs.
"""
import numpy as np
import pytest
import torch

from oracle import shared_attn_oracle as O

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
LINE_KERNELS = ("lines64", "lines32", "lines32k128", "lines64k128", "lines32k256")   # IR_PROBS_LINES*: rows x keys per wave and step


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


CASES = [
    # B, H, Lq, Ls, N, Lr, include_self        (line kernel needs Ls, Lr multiples of 8)
    (2, 2, 72, 72, 3, 40, True),       # ragged 64-key steps, partial row block
    (2, 2, 72, 72, 3, 40, False),
    (1, 3, 256, 256, 4, 256, True),    # the 16x16-token class, thin
    (1, 2, 300, 304, 2, 136, True),    # rows not a multiple of 32, key tails of 8 / 48 keys
    (1, 1, 520, 520, 1, 1032, False),  # more than one 256-row workgroup, a reference longer than the query axis
    (2, 1, 64, 64, 8, 64, True),       # eight references
    (1, 2, 96, 96, 0, 0, True),        # no references: plain self attention
    (1, 1, 1024, 1024, 4, 1024, True), # the 32x32-token class, one head: key chunks cut the segments
]
ODD_CASES = [
    (2, 2, 33, 33, 2, 37, True),       # nothing aligned: rows of P start at odd byte offsets -> generic kernel
    (1, 2, 64, 64, 3, 20, False),      # Lr % 8 != 0
    (1, 1, 77, 77, 1, 1, True),
]


def _case_inputs(case, dtype, seed=5):
    B, H, Lq, Ls, N, Lr, inc = case
    C = H * 64
    gen = torch.Generator().manual_seed(seed)
    q = _rand((B, Lq, C), dtype, gen, 1.5)
    k, v = _rand((B, Ls, C), dtype, gen, 1.5), _rand((B, Ls, C), dtype, gen)
    rk = _rand((B, N, Lr, C), dtype, gen, 1.5) if N else None
    rv = _rand((B, N, Lr, C), dtype, gen) if N else None
    return q, k, v, rk, rv


def _oracle_probs(q, k, v, rk, rv, H, inc):
    n = lambda t: None if t is None else _np64(t)
    _, p = O.shared_attention_np(_np64(q), _np64(k), _np64(v), n(rk), n(rv), H, 0.125, False, inc, return_probs=True)
    return p


def _gpu_lse(ops, q, k, v, rk, rv, H, inc):
    d = lambda t: None if t is None else t.cuda()
    qd, kd, vd, rkd, rvd = d(q), d(k), d(v), d(rk), d(rv)
    _, lse = ops.shared_attention(qd, kd, vd, rkd, rvd, heads=H, scale=0.125, include_self=inc, return_lse=True)
    return qd, kd, rkd, lse


def _ids(cases):
    return [f"B{c[0]}H{c[1]}L{c[2]}Ls{c[3]}N{c[4]}Lr{c[5]}s{int(c[6])}" for c in cases]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=_ids(CASES))
def test_every_probs_kernel_against_the_oracle(ops, case, dtype):
    B, H, Lq, Ls, N, Lr, inc = case
    q, k, v, rk, rv = _case_inputs(case, dtype)
    p_ref = _oracle_probs(q, k, v, rk, rv, H, inc)
    qd, kd, rkd, lse = _gpu_lse(ops, q, k, v, rk, rv, H, inc)
    got = {}
    for kern in ("auto", "generic") + LINE_KERNELS:
        probs = ops.attn_probs(qd, kd, rkd, lse, heads=H, scale=0.125, include_self=inc, kernel=kern)
        assert probs.shape == p_ref.shape and probs.dtype == dtype
        p = probs.float().cpu().numpy()
        assert np.isfinite(p).all(), kern
        assert np.abs(p - p_ref).max() <= TOL[dtype], (kern, np.abs(p - p_ref).max())
        np.testing.assert_allclose(p.sum(-1), 1.0, atol=4 * TOL[dtype])
        got[kern] = probs
    for kern in ("auto",) + LINE_KERNELS:
        assert torch.equal(got[kern], got["generic"]), f"{kern} differs from the 2-byte-store kernel"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("case", ODD_CASES, ids=_ids(ODD_CASES))
def test_unaligned_lengths_take_the_generic_kernel(ops, case, dtype):
    B, H, Lq, Ls, N, Lr, inc = case
    q, k, v, rk, rv = _case_inputs(case, dtype, seed=6)
    p_ref = _oracle_probs(q, k, v, rk, rv, H, inc)
    qd, kd, rkd, lse = _gpu_lse(ops, q, k, v, rk, rv, H, inc)
    probs = ops.attn_probs(qd, kd, rkd, lse, heads=H, scale=0.125, include_self=inc)
    assert np.abs(probs.float().cpu().numpy() - p_ref).max() <= TOL[dtype]
    for kern in LINE_KERNELS:   # asked for by name on a shape it does not cover: refused, not silently replaced
        with pytest.raises(ops._lib.IRError, match="multiples of 8"):
            ops.attn_probs(qd, kd, rkd, lse, heads=H, scale=0.125, include_self=inc, kernel=kern)


def test_rows_past_the_output_are_untouched(ops):
    """canary: the line kernel's range-checked stores must not write outside (B, H, Lq, Lkv) - the buffer behind it keeps its
    fill pattern (partial row block, key tail that is not a whole 64-key step)"""
    dtype = torch.float16
    case = (1, 2, 72, 72, 2, 40, True)
    B, H, Lq, Ls, N, Lr, inc = case
    q, k, v, rk, rv = _case_inputs(case, dtype)
    qd, kd, rkd, lse = _gpu_lse(ops, q, k, v, rk, rv, H, inc)
    from instantrestore_amd import _lib
    import ctypes as C
    lkv = Ls + N * Lr
    n = B * H * Lq * lkv
    big = torch.full((n + 65536,), 7.0, dtype=dtype, device="cuda")
    args, *_keep = ops._probs_args(qd, kd, rkd, lse, H, 0.125, inc)
    for kern in (2, 3, 4, 5, 6):
        big.fill_(7.0)
        _lib.check(_lib.lib().ir_attn_probs_ex(C.byref(args), big.data_ptr(), kern, torch.cuda.current_stream().cuda_stream), "probs")
        torch.cuda.synchronize()
        assert bool((big[n:] == 7.0).all()), "stores past the end of attention_probs"
        assert bool((big[:n] <= 1.0).all())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("case", CASES + ODD_CASES, ids=_ids(CASES + ODD_CASES))
def test_segment_mass(ops, case, dtype):
    """mass[b,h,i,s] = sum of row i's probabilities over segment s, fp32, summed before the 16-bit rounding: against the
    oracle's float64 block sums (abs 2e-3: the only 16-bit quantity left is the input rounding) and rows summing to 1"""
    B, H, Lq, Ls, N, Lr, inc = case
    q, k, v, rk, rv = _case_inputs(case, dtype, seed=7)
    p_ref = _oracle_probs(q, k, v, rk, rv, H, inc)
    if rk is None:
        inc = True
    edges = [0] + ([Ls] if inc else []) + [(Ls if inc else 0) + (n + 1) * Lr for n in range(N)]
    m_ref = np.stack([p_ref[..., a:b].sum(-1) for a, b in zip(edges[:-1], edges[1:])], axis=-1)
    qd, kd, rkd, lse = _gpu_lse(ops, q, k, v, rk, rv, H, inc)
    mass = ops.attn_segment_mass(qd, kd, rkd, lse, heads=H, scale=0.125, include_self=inc)
    assert mass.shape == m_ref.shape and mass.dtype == torch.float32
    m = mass.cpu().numpy()
    assert np.abs(m - m_ref).max() <= 2e-3, np.abs(m - m_ref).max()
    np.testing.assert_allclose(m.sum(-1), 1.0, atol=2e-3)


def test_processor_attention_mass_is_the_block_sum_of_attention_probs(ops):
    """opt-in ``SharedAttnProcessor.save_attention_mass``: the per-reference mass a gradio_demo.py:119-127-style consumer needs,
    equal to the block sums of the ``attention_probs`` the same call dumps (fp32 sums of the 16-bit probabilities: 4 x TOL)"""
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    torch.manual_seed(4)
    B, H, L, N = 2, 2, 256, 4
    C = H * 64
    for train_input in (True, False):
        proc = SharedAttnProcessor(self_attn_idx=0, save_self_attentions=True, use_adain=True, train_input=train_input)
        proc.save_attention_mass = True
        attn = Attention(query_dim=C, heads=H, dim_head=64, processor=proc).cuda()
        x = torch.randn(B, L, C, device="cuda")
        rk = torch.randn(B, N, L, C, device="cuda", dtype=torch.bfloat16)
        rv = torch.randn(B, N, L, C, device="cuda", dtype=torch.bfloat16)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            attn(x, ref_keys=[rk], ref_values=[rv])
        S = N + int(train_input)
        assert proc.attention_mass.shape == (B, H, L, S) and proc.attention_mass.dtype == torch.float32
        blocks = proc.attention_probs.float().reshape(B, H, L, S, L).sum(-1)
        assert float((proc.attention_mass - blocks).abs().max()) <= 4 * 8e-3
        assert float((proc.attention_mass.sum(-1) - 1).abs().max()) <= 2e-3
        proc.save_self_attentions = False          # the mass alone: no (B, H, L, Lkv) tensor is formed
        proc.attention_probs = None
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            attn(x, ref_keys=[rk], ref_values=[rv])
        assert proc.attention_probs is None and float((proc.attention_mass.sum(-1) - 1).abs().max()) <= 2e-3


def test_cfg5_top_layer_shape_one_head(ops):
    """the 1024 px layer class (L = 16 384 query rows, Lkv = 81 920 keys): one (b, h) of the probability matrix is 2.7 GB - the
    reference itself cannot form the cfg-5 tensor (215 GB at B = 16).  Rows sum to 1, sampled rows equal the oracle's, and the
    line kernel equals the 2-byte-store kernel on a row band (bit for bit)."""
    dtype = torch.float16
    B, H, L, N = 1, 1, 16384, 4
    g = torch.Generator(device="cuda").manual_seed(2)
    q = (torch.randn(B, L, 64, device="cuda", generator=g) * 1.2).to(dtype)
    k = torch.randn(B, L, 64, device="cuda", generator=g).to(dtype)
    v = torch.randn(B, L, 64, device="cuda", generator=g).to(dtype)
    rk = torch.randn(B, N, L, 64, device="cuda", generator=g).to(dtype)
    rv = torch.randn(B, N, L, 64, device="cuda", generator=g).to(dtype)
    _, lse = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=True, return_lse=True)
    probs = ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=True)
    assert probs.shape == (B, H, L, 5 * L)
    sums = probs.float().sum(-1)
    assert float((sums - 1).abs().max()) <= 4 * TOL[dtype]
    rows = sorted(set(np.random.default_rng(1).integers(0, L, 40).tolist() + [0, 63, 64, L - 1]))
    _, p_ref = O.shared_attention_np(_np64(q)[:, rows], _np64(k), _np64(v), _np64(rk), _np64(rv), H, 0.125, False, True, return_probs=True)
    assert np.abs(probs[:, :, rows].float().cpu().numpy() - p_ref).max() <= TOL[dtype]
    band = slice(4096, 4096 + 512)
    lse_b = lse[:, :, band].contiguous()
    a = ops.attn_probs(q[:, band].contiguous(), k, rk, lse_b, heads=H, scale=0.125, include_self=True, kernel="generic")
    assert torch.equal(a, probs[:, :, band])


def test_randomised_shapes_line_kernels_equal_the_generic_kernel(ops):
    """seeded sweep over ragged shapes (IR_SWEEP_CASES widens it): every line kernel must reproduce the 2-byte-store kernel bit
    for bit wherever it applies, rows sum to 1, and lengths that are not multiples of 8 must still work through `auto`"""
    import os
    rng = np.random.default_rng(int(os.environ.get("IR_SWEEP_SEED", "77")))
    gen = torch.Generator().manual_seed(77)
    for case in range(int(os.environ.get("IR_SWEEP_CASES", "24"))):
        B, H = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        unit = 8 if case % 4 else 1                       # every fourth case: lengths that are NOT multiples of 8
        Lq = int(rng.integers(1, 400))
        Ls = int(rng.integers(1, 60)) * unit
        N = int(rng.integers(0, 6))
        Lr = int(rng.integers(1, 50)) * unit if N else 0
        inc = bool(rng.integers(0, 2)) or N == 0
        dtype = [torch.float16, torch.bfloat16][case % 2]
        C = H * 64
        q = _rand((B, Lq, C), dtype, gen, 1.4).cuda()
        k, v = _rand((B, Ls, C), dtype, gen, 1.4).cuda(), _rand((B, Ls, C), dtype, gen).cuda()
        rk = _rand((B, N, Lr, C), dtype, gen, 1.4).cuda() if N else None
        rv = _rand((B, N, Lr, C), dtype, gen).cuda() if N else None
        what = f"case {case}: B{B} H{H} Lq{Lq} Ls{Ls} N{N} Lr{Lr} inc{inc} {dtype}"
        if inc and Ls != Lq:
            # self K/V of another length than the query axis: cross-attention through the same entry point
            pass
        _, lse = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=inc, return_lse=True)
        ref = ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=inc, kernel="generic")
        auto = ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=inc)
        assert torch.equal(auto, ref), what
        assert float((ref.float().sum(-1) - 1).abs().max()) <= 4 * TOL[dtype], what
        aligned = (not inc or Ls % 8 == 0) and (N == 0 or Lr % 8 == 0)
        for kern in LINE_KERNELS:
            if aligned:
                assert torch.equal(ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=inc, kernel=kern), ref), (what, kern)
        mass = ops.attn_segment_mass(q, k, rk, lse, heads=H, scale=0.125, include_self=inc)
        edges = [0] + ([Ls] if inc else []) + [(Ls if inc else 0) + (n + 1) * Lr for n in range(N)]
        blocks = torch.stack([ref[..., a:b].float().sum(-1) for a, b in zip(edges[:-1], edges[1:])], dim=-1)
        assert float((mass - blocks).abs().max()) <= 4 * TOL[dtype], what


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 2, 72, 3, 40), (1, 2, 300, 2, 136), (1, 1, 1024, 4, 1024)], ids=["L72", "L300", "L1024"])
def test_segment_mass_with_prescaled_q(ops, dtype, shape):
    """IR_FLAG_Q_PRESCALED on the dump entry points (q holds Q * scale * log2(e): its products with k ARE the exponents, `scale`
    stays the unit of the LSE): the same numbers as the unflagged call with scale = ln 2 on the same pre-scaled q, and the
    float64 block sums of softmax(q' k^T ln 2).  (An exponent-domain form of the mass kernel - minus the LSE through the MFMA
    C operand, no multiply-add per score - was built and measured SLOWER, 0.853 vs 0.810 ms at the cfg-2 top layer: the
    16-register C blocks cost a wave of occupancy; not kept, NOTES 11.1.)"""
    B, H, L, N, Lr = shape
    C = H * 64
    gen = torch.Generator().manual_seed(11)
    q = _rand((B, L, C), dtype, gen, 1.5)
    qs = (q.float() * (0.125 * 1.4426950408889634)).to(dtype).cuda()          # what the fused q/k/v projection hands over
    k, v = _rand((B, L, C), dtype, gen, 1.5).cuda(), _rand((B, L, C), dtype, gen).cuda()
    rk, rv = _rand((B, N, Lr, C), dtype, gen, 1.5).cuda(), _rand((B, N, Lr, C), dtype, gen).cuda()
    _, lse = ops.shared_attention(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=True, return_lse=True, q_prescaled=True)
    m_presc = ops.attn_segment_mass(qs, k, rk, lse, heads=H, scale=0.125, include_self=True, q_prescaled=True)
    m_plain = ops.attn_segment_mass(qs, k, rk, lse, heads=H, scale=0.6931471805599453, include_self=True)
    assert float((m_presc - m_plain).abs().max()) <= 1e-4
    _, p_ref = O.shared_attention_np(_np64(qs), _np64(k), _np64(v), _np64(rk), _np64(rv), H, 0.6931471805599453, False, True, return_probs=True)
    edges = [0, L] + [L + (n + 1) * Lr for n in range(N)]
    m_ref = np.stack([p_ref[..., a:b].sum(-1) for a, b in zip(edges[:-1], edges[1:])], axis=-1)
    assert np.abs(m_presc.cpu().numpy() - m_ref).max() <= 2e-3
    # and the dump itself accepts the flag (same expression with the factor 1)
    p1 = ops.attn_probs(qs, k, rk, lse, heads=H, scale=0.125, include_self=True, q_prescaled=True)
    assert np.abs(p1.float().cpu().numpy() - p_ref).max() <= TOL[dtype]
