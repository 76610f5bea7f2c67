"""GPU parity of ir_linear_fwd (the K <= 320 projection GEMM and its K = 640 split-contraction form) against a float64 matmul of the same
16-bit inputs.  Tolerance (floating point): fp32 accumulation + one rounding to the 16-bit output:
|err| <= 2^-10 (fp16) / 2^-7 (bf16) relative to max(1, |y|) - half an output ulp plus slack for the
accumulation order."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("M,N,K,bias", [
    (256, 32, 64, False), (300, 96, 128, True), (1000, 960, 320, False), (4096, 320, 320, True),
    (33, 64, 320, True), (1, 32, 64, False), (8192, 960, 320, False), (777, 1920, 256, True), (512, 3840, 192, False),
    # K = 640: the contraction split over two waves (partials meet in LDS)
    (256, 32, 640, False), (300, 96, 640, True), (1, 64, 640, True), (8192, 1920, 640, False), (8192, 640, 640, True),
    (2049, 704, 640, True), (32768, 640, 640, True),
])
def test_linear_matches_float64(dtype, M, N, K, bias):
    from instantrestore_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g).to(dtype) if bias else None
    assert ops.linear_supported(x.cuda(), w.cuda(), None if b is None else b.cuda())
    y = ops.linear(x.cuda(), w.cuda(), None if b is None else b.cuda())
    assert y.shape == (M, N) and y.dtype == dtype
    ref = x.double().numpy() @ w.double().numpy().T
    if bias:
        ref = ref + b.double().numpy()
    err = np.abs(y.double().cpu().numpy() - ref)
    bound = TOL[dtype] * np.maximum(1.0, np.abs(ref))
    assert (err <= bound).all(), (err.max(), np.unravel_index(np.argmax(err - bound), err.shape))


def test_linear_exact_column_order_and_strided_input():
    """integer-valued inputs make every product exact: any permutation of output columns / rows or a
    wrong lane-pair exchange shows up as an exact mismatch.  Also batched (B, L, K) input views."""
    from instantrestore_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-3, 4, (2, 130, 320), generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (96, 320), generator=g).to(torch.bfloat16)
    b = torch.randint(-4, 5, (96,), generator=g).to(torch.bfloat16)
    y = ops.linear(x.cuda(), w.cuda(), b.cuda()).cpu()
    ref = (x.float() @ w.float().T + b.float())
    assert y.shape == (2, 130, 96)
    assert torch.equal(y.float(), ref.to(torch.bfloat16).float())
    # weight given as a row slice of a bigger (fused) weight: w_ld > K is not needed, but N offset is
    wbig = torch.randint(-2, 3, (3 * 96, 320), generator=g).to(torch.bfloat16).cuda()
    y2 = ops.linear(x.cuda(), wbig[96:192]).cpu()
    assert torch.equal(y2.float(), (x.float() @ wbig[96:192].cpu().float().T).to(torch.bfloat16).float())
    # the same with K = 640 (two K halves in two waves: a swapped or dropped half is an exact mismatch)
    x6 = torch.randint(-3, 4, (2, 130, 640), generator=g).to(torch.bfloat16)
    w6 = torch.randint(-2, 3, (96, 640), generator=g).to(torch.bfloat16)
    y6 = ops.linear(x6.cuda(), w6.cuda(), b.cuda()).cpu()
    assert torch.equal(y6.float(), (x6.float() @ w6.float().T + b.float()).to(torch.bfloat16).float())


TILED = ["256x128", "128x128", "128x64", "256x64", "64x128", "128x256", "256x256", "128x128k2"]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("kernel", TILED)
@pytest.mark.parametrize("M,N,K,bias", [
    (256, 128, 64, False), (300, 256, 128, True), (2048, 1280, 1280, True), (2048, 3840, 1280, False),
    (8192, 1280, 1280, True), (1, 128, 1280, True), (257, 384, 1280, False), (8192, 640, 640, True), (4099, 1920, 640, False),
    (32768, 320, 320, True), (129, 64, 320, True),
])
def test_tiled_linear_matches_float64(dtype, kernel, M, N, K, bias):
    """the LDS-tiled kernel (round 3: K = 1280 and the small-M shapes, every tile shape) against a float64 product of the
    same 16-bit inputs; with fp32 activations the result must be the SAME BITS as with the pre-cast ones (the fused cast is
    `.to(dtype)`)"""
    from instantrestore_amd import ops
    if N % (64 if kernel.startswith("256x256") else int(kernel.split("x")[1].split("k")[0])) != 0:   # 256x256: ragged last column tile
        pytest.skip("tile width does not divide N")
    if kernel.endswith("k2") and (K // 64) % 2:
        pytest.skip("the split-K tile takes an even number of 64-wide K steps")
    kid = ops.LIN_KERNELS[kernel]
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x32 = torch.randn(M, K, generator=g)
    x = x32.to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g).to(dtype) if bias else None
    bc = None if b is None else b.cuda()
    y = ops.linear(x.cuda(), w.cuda(), bc, kernel=kid)
    assert y.shape == (M, N) and y.dtype == dtype
    ref = x.double().numpy() @ w.double().numpy().T
    if bias:
        ref = ref + b.double().numpy()
    err = np.abs(y.double().cpu().numpy() - ref)
    bound = TOL[dtype] * np.maximum(1.0, np.abs(ref))
    assert (err <= bound).all(), (err.max(), np.unravel_index(np.argmax(err - bound), err.shape))
    y32 = ops.linear(x32.cuda(), w.cuda(), bc, kernel=kid)
    assert torch.equal(y32, y)
    # column scale of the pre-scaled-Q contract: columns [0, 64) times 0.18 in fp32 before the one rounding
    ys = ops.linear(x.cuda(), w.cuda(), bc, kernel=kid, scale_cols=64, col_scale=0.18)
    refs = ref.copy()
    refs[:, :64] = (refs[:, :64] - (b.double().numpy()[:64] if bias else 0.0)) * np.float32(0.18) + (b.double().numpy()[:64] if bias else 0.0)
    errs = np.abs(ys.double().cpu().numpy() - refs)
    assert (errs <= TOL[dtype] * np.maximum(1.0, np.abs(refs))).all()
    assert torch.equal(ys[:, 64:], y[:, 64:])


@pytest.mark.parametrize("kernel", TILED)
def test_tiled_linear_exact_layout(kernel):
    """integer-valued operands make every product exact: a permuted output column / row, a wrong swizzle slot or a stale
    LDS stage is an exact mismatch.  Ragged M, strided x rows, a row slice of a fused weight."""
    from instantrestore_amd import ops
    kid = ops.LIN_KERNELS[kernel]
    g = torch.Generator().manual_seed(11)
    xbig = torch.randint(-3, 4, (3, 333, 1280 + 64), generator=g).to(torch.bfloat16).cuda()
    x = xbig[..., :1280]                       # row stride 1344 elements
    wbig = torch.randint(-2, 3, (3 * 256, 1280), generator=g).to(torch.bfloat16).cuda()
    b = torch.randint(-4, 5, (256,), generator=g).to(torch.bfloat16).cuda()
    y = ops.linear(x, wbig[256:512], b, kernel=kid)
    ref = (x.float() @ wbig[256:512].float().T + b.float())
    assert y.shape == (3, 333, 256)
    assert torch.equal(y.float(), ref.to(torch.bfloat16).float())


@pytest.mark.parametrize("M,N,K,bias", [(2048, 1280, 1280, True), (2048, 3840, 1280, False), (333, 256, 640, True)])
def test_split_k_tile_is_bit_stable(M, N, K, bias):
    """the split-K tile (round 5) adds its two K halves in ONE fixed order through LDS - no atomics, nothing through memory:
    50 launches, fp32 and 16-bit activations, the same bits every time (and the fused cast is still `.to(dtype)`)"""
    from instantrestore_amd import ops
    g = torch.Generator().manual_seed(M + N)
    x32 = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).to(torch.bfloat16).cuda() if bias else None
    kid = ops.LIN_KERNELS["128x128k2"]
    first = ops.linear(x32, w, b, kernel=kid)
    assert torch.equal(first, ops.linear(x32.to(torch.bfloat16), w, b, kernel=kid))
    for _ in range(50):
        assert torch.equal(ops.linear(x32, w, b, kernel=kid), first)
    # against the single-pass tile: same product, another fp32 summation order - within one 16-bit rounding of each other
    ref = ops.linear(x32, w, b, kernel=ops.LIN_KERNELS["128x128"])
    assert float((first.float() - ref.float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref.float().abs().max()))


def test_linear_auto_choice_covers_every_projection_of_the_topology():
    """no projection of the SD-Turbo attention topology is left to a vendor GEMM"""
    from instantrestore_amd import ops
    for sets in (1, 4, 8, 32, 64):
        for (L, C) in ((256, 1280), (1024, 640), (4096, 320), (16384, 320)):
            for N, bias in ((3 * C, False), (C, True), (2 * C, False)):
                assert ops.linear_kernel_for(sets * L, N, C, bias) >= 1, (sets, L, C, N)
    assert ops.linear_kernel_for(77 * 8, 1280, 1024, False) >= 2      # cross-attention to_k/to_v: text width 1024


def test_linear_rejects_what_it_does_not_implement():
    from instantrestore_amd import _lib, ops
    x = torch.zeros(64, 1000, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(64, 1000, device="cuda", dtype=torch.bfloat16)
    assert not ops.linear_supported(x, w, None)
    with pytest.raises(_lib.IRError):
        ops.linear(x, w)
    x = torch.zeros(64, 1280, device="cuda", dtype=torch.bfloat16)
    w = torch.zeros(96, 1280, device="cuda", dtype=torch.bfloat16)      # K = 1280 needs N % 64 == 0
    assert not ops.linear_supported(x, w, None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(32, 64, dtype=torch.bfloat16))


def test_linear_stress_full_size_against_vendor_gemm():
    """the kernel keeps W chunks and result stores in flight across barriers and waits on COUNTED
    s_waitcnt values: a miscount would show up as sporadically stale chunks.  Full top-layer size,
    many launches back to back, every result compared with the vendor GEMM's."""
    from instantrestore_amd import ops
    torch.manual_seed(0)
    w = (torch.randn(960, 320, device="cuda") / 18.0).to(torch.bfloat16)
    wo = (torch.randn(320, 320, device="cuda") / 18.0).to(torch.bfloat16)
    bo = torch.randn(320, device="cuda").to(torch.bfloat16)
    for it in range(12):
        x = torch.randn(131072 - 37 * it, 320, device="cuda").to(torch.bfloat16)   # ragged last row block too
        y, ref = ops.linear(x, w), torch.nn.functional.linear(x, w)
        assert (y.float() - ref.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, ref.float().abs().max().item()), it
        y2, ref2 = ops.linear(x, wo, bo), torch.nn.functional.linear(x, wo, bo)
        assert (y2.float() - ref2.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, ref2.float().abs().max().item()), it
    # the K = 640 form: partial tiles and staged stores share LDS regions across chunk parities
    w6 = (torch.randn(1920, 640, device="cuda") / 25.0).to(torch.bfloat16)
    wo6 = (torch.randn(640, 640, device="cuda") / 25.0).to(torch.bfloat16)
    bo6 = torch.randn(640, device="cuda").to(torch.bfloat16)
    for it in range(12):
        x = torch.randn(32768 - 37 * it, 640, device="cuda").to(torch.bfloat16)
        y, ref = ops.linear(x, w6), torch.nn.functional.linear(x, w6)
        assert (y.float() - ref.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, ref.float().abs().max().item()), it
        y2, ref2 = ops.linear(x, wo6, bo6), torch.nn.functional.linear(x, wo6, bo6)
        assert (y2.float() - ref2.float()).abs().max().item() <= 2.0 ** -6 * max(1.0, ref2.float().abs().max().item()), it


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("k", [320, 640])
def test_leading_columns_scaled_before_the_rounding(dtype, k):
    """ir_linear_fwd_scaled: columns [0, scale_cols) carry col_scale, applied to the fp32 accumulator (one rounding):
    equal to rounding the scaled float64 product, NOT to scaling the rounded product"""
    from instantrestore_amd import ops
    torch.manual_seed(4)
    m, n, sc, cs = 1000, 3 * k, k, 0.125 * 1.4426950408889634
    x = torch.randn(m, k).to(dtype)
    w = (torch.randn(n, k) / k ** 0.5).to(dtype)
    y = ops.linear(x.cuda(), w.cuda(), None, scale_cols=sc, col_scale=cs).float().cpu().double()
    ref = x.double() @ w.double().T
    ref[:, :sc] *= cs
    ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[dtype]
    assert (y - ref).abs().max().item() <= ulp * 1.05 * max(1.0, ref.abs().max().item())   # half an ulp of the largest value + accumulation
    plain = ops.linear(x.cuda(), w.cuda(), None).float().cpu().double()
    assert torch.equal(plain[:, sc:], y[:, sc:])                                             # the other columns are untouched


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(20000, 960, 320), (3000, 192, 128), (9000, 1920, 640), (777, 640, 640)])
def test_fp32_activations_are_cast_while_loaded(dtype, shape):
    """x in fp32 (LayerNorm output under autocast): ir_linear_fwd_scaled(x_is_f32) must give the bytes of
    linear(x.to(dtype)) through the same kernel - the fused cast is the same round-to-nearest-even"""
    from instantrestore_amd import ops
    m, n, k = shape
    torch.manual_seed(m)
    x = (torch.randn(m, k) * 3.0).cuda()
    x[0, :8] = torch.tensor([1.0 + 2 ** -9, 1.0 + 2 ** -8, -1.0 - 3 * 2 ** -10, 65504.0, 1e-8, -0.0, 3.3e4, 2 ** -15]).cuda()
    w = (torch.randn(n, k) / k ** 0.5).to(dtype).cuda()
    bias = torch.randn(n).to(dtype).cuda() if n <= 640 else None
    a = ops.linear(x, w, bias)
    b = ops.linear(x.to(dtype), w, bias)
    assert a.dtype == dtype and torch.equal(a, b)


def test_randomised_projection_shapes_every_kernel():
    """seeded sweep (IR_SWEEP_CASES / IR_SWEEP_SEED widen it): random M (whole tiles, ragged tails, one row), N, K over both kernel
    families, bias or not, 16-bit or fp32 activations, rows of X with a pitch larger than K, a scaled leading column range - through
    the automatic choice AND through every kernel that accepts the shape, against a float64 product of the same rounded operands;
    the statistics-tail form must return the same bits of Y where a shape can carry it"""
    import os
    from instantrestore_amd import ops
    seed = int(os.environ.get("IR_SWEEP_SEED", "404"))
    rng = np.random.default_rng(seed)
    g = torch.Generator(device="cuda").manual_seed(seed)
    ks = [32, 64, 96, 128, 160, 192, 256, 288, 320, 384, 512, 640, 704, 768, 1024, 1280, 1344]
    tried = {}
    for case in range(int(os.environ.get("IR_SWEEP_CASES", "40"))):
        dtype = [torch.bfloat16, torch.float16][case % 2]
        K = int(rng.choice(ks))
        N = 32 * int(rng.integers(1, 41)) if rng.integers(0, 3) else 64 * int(rng.integers(1, 61))
        mode = int(rng.integers(0, 4))
        M = [int(rng.integers(1, 70)), int(rng.integers(1, 3000)), 64 * int(rng.integers(1, 80)), 256 * int(rng.integers(1, 40)) + int(rng.integers(0, 2))][mode]
        bias = bool(rng.integers(0, 2))
        f32 = bool(rng.integers(0, 2))
        pitch = K + 8 * int(rng.integers(0, 4))
        xbuf = torch.randn(M, pitch, device="cuda", generator=g)
        x32 = xbuf[:, :K]
        x16 = x32.to(dtype)
        x = x32 if f32 else (torch.empty(M, pitch, dtype=dtype, device="cuda")[:, :K].copy_(x16))
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(dtype)
        b = torch.randn(N, device="cuda", generator=g).to(dtype) if bias else None
        sc = 32 * int(rng.integers(0, N // 32 + 1)) if rng.integers(0, 2) else 0
        what = f"case {case}: M{M} N{N} K{K} bias{bias} f32{f32} pitch{pitch} scale_cols{sc} {dtype}"
        if not ops.linear_supported(x, w, b):
            continue
        acc = x16.double() @ w.double().T
        if sc:
            acc[:, :sc] = acc[:, :sc] * float(np.float32(0.18))     # the fp32 accumulator times the fp32 factor, before bias and rounding
        ref = acc + (b.double() if bias else 0.0)
        bound = TOL[dtype] * torch.clamp(ref.abs(), min=1.0)
        auto = None
        for name, kid in ops.LIN_KERNELS.items():
            try:
                y = ops.linear(x, w, b, kernel=kid, scale_cols=sc, col_scale=0.18)
            except RuntimeError:
                assert kid != 0, what                      # the automatic choice must serve what linear_supported accepted
                continue
            tried[name] = tried.get(name, 0) + 1
            assert y.shape == (M, N) and y.dtype == dtype, (what, name)
            err = (y.double() - ref).abs()
            assert bool((err <= bound).all()), (what, name, float((err - bound).max()))
            if kid == 0:
                auto = y
        rows = ops.linear_stats_rows(M, N, K, bias)
        if rows > 0 and N >= 64:
            heads = N // 64
            h0 = int(rng.integers(0, heads))
            hc = int(rng.integers(1, heads - h0 + 1))
            ys, st = ops.linear(x, w, b, scale_cols=sc, col_scale=0.18, stats=(64 * h0, 64 * hc))
            assert torch.equal(ys, auto), what
            # the partials of the column range merge into the token mean of those columns (one set of M rows)
            if M // rows <= ops.STATS_MAX_CHUNKS:
                mean, std = ops.token_stats_from_partials(st, 1, M)
                cols = ys[:, 64 * h0:64 * (h0 + hc)].double()
                assert float((mean.reshape(-1).double() - cols.mean(0)).abs().max()) <= 1e-3 * max(1.0, float(cols.abs().max())), what
                if M > 1:
                    assert float((std.reshape(-1).double() - cols.std(0)).abs().max()) <= 2e-3 * max(1.0, float(cols.abs().max())), what
    print("linear sweep: launches per kernel", tried)
