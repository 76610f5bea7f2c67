"""The 128-rows-per-wave attention kernel (csrc/shared_attn_fwd_w128.hip, IR_TUNE_W128 = 16; round 6) against the oracle: every
shape class it covers - pre-scaled Q, whole 64-key tiles - with and without the AdaIN fold, with and without the self block, both
dtypes, query axes that are not multiples of its 512-row items, item grids with and without the K/V-range split of the remainder
round, one to eight references, and the configurations' own batches.  Bounds: tests/parity_bounds.py (stated + regression).
The instruction stream itself is also executed instruction by instruction on the CPU (tests/test_w128_stream.py)."""
import numpy as np
import pytest
import torch

from oracle import shared_attn_oracle as O
from parity_bounds import check_before_rounding, check_parity

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634
C_ = 0.125 * LOG2E
W128 = 16


def _inputs(B, H, Lq, Ls, N, Lr, dtype, seed, peaky=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, device="cuda")
    C = H * 64
    sc = 2.5 if peaky else 1.0
    q = (rnd(B, Lq, C) * sc).to(dtype)
    k, v = (rnd(B, Ls, C) * sc).to(dtype), (rnd(B, Ls, C) * 0.9 + 0.3).to(dtype)
    rk = (rnd(B, N, Lr, C) * sc).to(dtype) if N else None
    rv = (rnd(B, N, Lr, C) * 1.4 - 0.2).to(dtype) if N else None
    qs = (q.float() * C_).to(dtype)                                    # what the fused projection hands over (one rounding)
    return qs, qs.float() / C_, k, v, rk, rv


def _run(ops, qs, k, v, rk, rv, H, inc, aff, variant=W128, **kw):
    ops.set_attn_variant(variant)
    try:
        return ops.shared_attention(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=inc, adain=aff, q_prescaled=True, **kw)
    finally:
        ops.set_attn_variant(0)


CASES = [
    # B, H, Lq, Ls, N, Lr, include_self, adain, peaky
    (1, 1, 128, 128, 0, 0, True, False, False),       # one wave's worth of rows, two tiles, plain self-attention
    (1, 2, 512, 512, 4, 512, True, True, False),      # one full item per head
    (2, 2, 64, 64, 2, 64, True, True, False),         # fewer rows than one wave holds, one-tile segments (runs of one tile)
    (1, 1, 576, 576, 3, 192, False, True, True),      # query axis not a multiple of 512; no self block; three-tile references
    (1, 3, 1024, 1024, 1, 64, True, False, True),
    (2, 1, 640, 128, 8, 128, True, True, False),      # eight references, Ls != Lq
    (1, 5, 4096, 4096, 4, 4096, True, True, False),   # cfg 2's top layer, one identity: 40 items, unsplit
    (3, 5, 4096, 4096, 4, 4096, True, True, False),   # 120 items on 256 slots
    (7, 5, 4096, 4096, 2, 4096, False, True, False),  # 280 items: 24 items in the remainder round -> K/V-range pieces + combine
    (1, 5, 4608, 4096, 4, 4096, True, False, False),  # 45 items (nqb = 9)
    (1, 2, 192, 128, 17, 64, True, True, False),      # 18 segments: more than the LDS fold table holds (coefficients from global memory)
    (1, 1, 128, 64, 15, 64, True, True, False),       # 16 segments: the table's last slot
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("case", CASES, ids=[f"B{c[0]}H{c[1]}Lq{c[2]}Ls{c[3]}N{c[4]}Lr{c[5]}s{int(c[6])}a{int(c[7])}p{int(c[8])}" for c in CASES])
def test_w128_parity(case, dtype):
    from instantrestore_amd import ops
    B, H, Lq, Ls, N, Lr, inc, ad, peaky = case
    qs, q_eff, k, v, rk, rv = _inputs(B, H, Lq, Ls, N, Lr, dtype, 100 + Lq + 7 * N + B, peaky)
    aff = ops.adain_stats(v, rv, heads=H) if ad else None
    out, lse = _run(ops, qs, k, v, rk, rv, H, inc, aff, return_lse=True)
    name = ops.shared_attention_kernel_name(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=inc, adain=aff, q_prescaled=True)
    rows = torch.unique(torch.tensor([r % Lq for r in (0, 1, 31, 32, 63, 64, 127, 128, 255, 511, 512, Lq // 2, Lq - 2, Lq - 1)]
                                     + np.random.default_rng(Lq).integers(0, Lq, 24).tolist()))
    f = lambda t: None if t is None else t.float().cpu()
    for b in sorted({0, B - 1}):
        ref = O.shared_attention_port(q_eff[b:b + 1, rows.cuda()].cpu(), f(k[b:b + 1]), f(v[b:b + 1]),
                                      None if rk is None else f(rk[b:b + 1]), None if rv is None else f(rv[b:b + 1]), H, 0.125,
                                      use_adain=ad, train_input=inc)
        check_parity(out[b:b + 1, rows.cuda()], ref.numpy(), dtype, f"w128 {case} identity {b}", reg_factor=1.5 if peaky else 1.0)
    # the 64-row kernel on the same call: same result within both kernels' error, LSE equal
    out64, lse64 = _run(ops, qs, k, v, rk, rv, H, inc, aff, variant=13, return_lse=True)
    assert float((out.float() - out64.float()).abs().max()) <= 2 * (8e-3 if dtype == torch.bfloat16 else 1e-3) * max(1.0, float(out64.float().abs().max()))
    assert float((lse - lse64).abs().max()) <= 2e-3 * max(1.0, float(lse64.abs().max()))
    # bit-identical from launch to launch
    out2 = _run(ops, qs, k, v, rk, rv, H, inc, aff)
    assert torch.equal(out, out2)
    # the default dispatch reports the kernel it would take; the forced variant ran the 128-row kernel
    ops.set_attn_variant(W128)
    try:
        forced = ops.shared_attention_kernel_name(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=inc, adain=aff, q_prescaled=True)
    finally:
        ops.set_attn_variant(0)
    assert "w128" in forced, (forced, name)


@pytest.mark.parametrize("cfg,B,N,L,H,dtype", [("cfg2", 8, 4, 4096, 5, torch.bfloat16), ("cfg4", 8, 8, 4096, 5, torch.bfloat16),
                                               ("cfg5", 16, 4, 16384, 5, torch.float16)], ids=["cfg2", "cfg4", "cfg5"])
@pytest.mark.parametrize("train_input", [True, False], ids=["t1", "t0"])
def test_w128_at_the_configs_batch(cfg, B, N, L, H, dtype, train_input):
    """the top layer class of cfg 2 / 4 / 5 with the REAL grid (320 / 320 / 2560 items: whole rounds + the split remainder), shared
    form (fold) and capture form (plain self-attention over the B * N reference token sets); first and last identity"""
    from instantrestore_amd import ops
    qs, q_eff, k, v, rk, rv = _inputs(B, H, L, L, N, L, dtype, 1000 + L + N)
    aff = ops.adain_stats(v, rv, heads=H)
    out = _run(ops, qs, k, v, rk, rv, H, train_input, aff)
    rows = torch.tensor(sorted({0, 1, 31, 32, 63, 64, 127, 128, 255, 256, 511, 512, (L // 2 + 77) % L, L - 2, L - 1}))
    f = lambda t: t.float().cpu()
    refs = []
    for b in (0, B - 1):
        ref = O.shared_attention_port(q_eff[b:b + 1, rows.cuda()].cpu(), f(k[b:b + 1]), f(v[b:b + 1]), f(rk[b:b + 1]), f(rv[b:b + 1]), H,
                                      0.125, use_adain=True, train_input=train_input)
        check_parity(out[b:b + 1, rows.cuda()], ref.numpy(), dtype, f"{cfg} shared L={L} identity {b} (w128)")
        refs.append(ref)
    assert torch.equal(out, _run(ops, qs, k, v, rk, rv, H, train_input, aff))
    if dtype == torch.bfloat16:
        out32 = _run(ops, qs, k, v, rk, rv, H, train_input, aff, out_dtype=torch.float32)
        for b, ref in zip((0, B - 1), refs):
            check_before_rounding(out32[b:b + 1, rows.cuda()], ref.numpy(), f"{cfg} shared L={L} identity {b} fp32 out (w128)")
        assert torch.equal(out32.to(dtype), out)
    if train_input:
        # capture form: the B * N reference token sets as a batch of plain self-attentions
        S = B * N
        kk, vv = rk.reshape(S, L, H * 64), rv.reshape(S, L, H * 64)
        g = torch.Generator(device="cuda").manual_seed(3)
        qq = (torch.randn(S, L, H * 64, generator=g, device="cuda") * C_).to(dtype)
        oc = _run(ops, qq, kk, vv, None, None, H, True, None)
        for s_ in (0, S - 1):
            ref = O.shared_attention_port((qq.float() / C_)[s_:s_ + 1, rows.cuda()].cpu(), f(kk[s_:s_ + 1]), f(vv[s_:s_ + 1]), None, None, H, 0.125)
            check_parity(oc[s_:s_ + 1, rows.cuda()], ref.numpy(), dtype, f"{cfg} capture L={L} token set {s_} (w128)")


def test_w128_refuses_what_it_does_not_cover():
    from instantrestore_amd import ops
    qs, _, k, v, rk, rv = _inputs(1, 1, 200, 200, 2, 72, torch.bfloat16, 5)
    with pytest.raises(RuntimeError):
        _run(ops, qs, k, v, rk, rv, 1, True, None)                      # ragged segments
    qs, _, k, v, rk, rv = _inputs(1, 1, 128, 128, 2, 64, torch.bfloat16, 5)
    ops.set_attn_variant(W128)
    try:
        with pytest.raises(RuntimeError):
            ops.shared_attention(qs, k, v, rk, rv, heads=1, scale=0.125, include_self=True)   # Q not pre-scaled
    finally:
        ops.set_attn_variant(0)


def test_w128_randomised_sweep():
    """seeded random shapes inside the 128-row kernel's domain: batch x heads from 1 to 12 items and beyond a round of 256, query axes
    from 64 to 5 000 rows (not multiples of anything), self / reference lengths of 1-40 tiles, 0-6 references, both flags, both
    dtypes, with and without the workspace (K/V-range pieces on / off), peaky inputs every fourth case; sampled rows against the
    fp32 port.  IR_SWEEP_CASES / IR_SWEEP_SEED widen it for a soak (round 6: 400 cases x 3 seeds clean)."""
    import os
    from instantrestore_amd import ops
    seed = int(os.environ.get("IR_SWEEP_SEED", "606"))
    rng = np.random.default_rng(seed)
    ncase = int(os.environ.get("IR_SWEEP_CASES", "24"))
    for case in range(ncase):
        dtype = [torch.bfloat16, torch.float16][case % 2]
        B, H = int(rng.integers(1, 5)), int(rng.integers(1, 6))
        big = case % 6 == 5
        Lq = int(rng.integers(4000, 5000)) if big else int(rng.choice([64, 128, 200, 512, 576, 1000, 1024, 1300]))
        N = int(rng.integers(0, 7))
        inc = bool(rng.integers(0, 2)) or N == 0
        Ls = 64 * int(rng.integers(1, 41)) if inc else 64
        Lr = 64 * int(rng.integers(1, 41 if not big else 17)) if N else 0
        ad = bool(rng.integers(0, 2)) and N > 0
        peaky = case % 4 == 3
        split = bool(rng.integers(0, 2))
        qs, q_eff, k, v, rk, rv = _inputs(B, H, Lq, Ls, N, Lr, dtype, seed * 1000 + case, peaky)
        aff = ops.adain_stats(v, rv, heads=H) if ad else None
        what = f"sweep {case}: B{B} H{H} Lq{Lq} Ls{Ls} N{N} Lr{Lr} inc{inc} ad{ad} peaky{peaky} split{split} {str(dtype)[6:]}"
        out = _run(ops, qs, k, v, rk, rv, H, inc, aff, split=split)
        rows = torch.unique(torch.tensor([0, Lq // 2, Lq - 1] + rng.integers(0, Lq, 13).tolist()))
        f = lambda t: None if t is None else t.float().cpu()
        b = int(rng.integers(0, B))
        ref = O.shared_attention_port(q_eff[b:b + 1, rows.cuda()].cpu(), f(k[b:b + 1]), f(v[b:b + 1]),
                                      None if rk is None else f(rk[b:b + 1]), None if rv is None else f(rv[b:b + 1]), H, 0.125,
                                      use_adain=ad, train_input=inc)
        check_parity(out[b:b + 1, rows.cuda()], ref.numpy(), dtype, what, reg_factor=1.5 if peaky else 1.0)
