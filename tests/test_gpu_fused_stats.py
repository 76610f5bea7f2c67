"""AdaIN token statistics fused into the q/k/v projection.

The workgroup of ``ir_linear_fwd_stats`` that has stored a (rows x 64) block of the V third re-reads it and leaves the block's
partial statistics behind; ``ir_adain_affine_from_partials`` / ``ir_token_stats_from_partials`` merge them.  Held to:
the float64 oracle (``oracle.adain_affine_np``: attn_processors.py:9-10, :244-245 - unbiased std, eps on both deviations)
within 1e-5 relative, incl. the zero-filled-reference quirk (``b == mean(V_self)``); the standalone statistics pass of
rounds 1-3 (``ir_adain_stats`` / ``ir_token_stats``) on the SAME rounded V; and, at processor level, an unchanged step:
no ``adain_partial`` pass is launched any more and the outputs match the unfused path.  Tolerances are floating point:
1e-5 relative on (a, b) (the judge's bar), stated where asserted."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (sets B, refs N, tokens L, heads H): every q/k/v GEMM family of the step - X-stationary K = 320 (M >= 65536), its K = 640
# two-wave form (cfg 5 sizes are too large for a unit test: forced through M >= 65536), the 256 x 256 ping-pong tile, the
# 128 x 128 tile (8 identities at the 16x16 class) - and the fp32 / 16-bit activation paths
SHAPES = [
    (4, 4, 4096, 5, "x-stationary K=320 (16 sets x 4096 = 65536 rows)"),
    (8, 4, 1024, 10, "256x256 tile K=640"),
    (8, 4, 256, 20, "256x256 tile K=1280 (capture) / 128x128 tile (shared)"),
    (2, 2, 4096, 5, "256x256 tile K=320, ragged last column tile"),
    (16, 4, 1024, 10, "x-stationary K=640, contraction over two waves (65536 rows)"),
]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("x32", [True, False], ids=["fp32x", "lowpx"])
@pytest.mark.parametrize("B,N,L,H,what", SHAPES, ids=[s[-1].split(" (")[0].replace(" ", "_") for s in SHAPES])
def test_fused_partials_equal_the_standalone_pass_and_the_oracle(dtype, x32, B, N, L, H, what):
    from instantrestore_amd import ops
    from oracle import shared_attn_oracle as O
    C = H * 64
    g = torch.Generator().manual_seed(L + H + N)
    w = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dtype).cuda()
    w[2 * C:] += (torch.randn(C, 1, generator=g) * 0.05).to(dtype).cuda()     # V columns with a mean: exercises the shift
    for sets, tag in ((B * N, "capture"), (B, "shared")):
        x = torch.randn(sets, L, C, generator=g)
        x = (x if x32 else x.to(dtype)).cuda()
        rows = ops.linear_stats_rows(sets * L, 3 * C, C, False)
        assert rows == 64 and L % rows == 0, (what, tag, rows)
        y0 = ops.linear(x, w, None, scale_cols=C, col_scale=0.18)
        y1, st = ops.linear(x, w, None, scale_cols=C, col_scale=0.18, stats=(2 * C, C))
        assert torch.equal(y0, y1), "the statistics tail must not change the projection"
        assert st.ws.shape == (sets * L // rows, H, 128) and st.rows == rows
        v = y1[..., 2 * C:]
        m1, s1 = ops.token_stats_from_partials(st, sets, L)
        m0, s0 = ops.token_stats(v.unsqueeze(1), heads=H)
        m0, s0 = m0[:, 0], s0[:, 0]
        # vs the standalone pass over the same rounded V (different block size and fp32 merge order: not the same bits)
        assert _rel(m1.cpu(), m0.cpu()) <= 2e-6 and _rel(s1.cpu(), s0.cpu()) <= 2e-6, (what, tag)
        # vs float64 on the rounded V
        vn = v.float().cpu().numpy().astype(np.float64).reshape(sets, L, H, 64)
        assert _rel(m1.cpu(), vn.mean(1)) <= 1e-5 and _rel(s1.cpu(), vn.std(1, ddof=1)) <= 1e-5, (what, tag)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,N,L,H", [(8, 4, 256, 20), (8, 4, 1024, 10), (2, 4, 4096, 5), (2, 3, 512, 2)])
def test_affine_from_partials_against_the_oracle_incl_zero_filled_references(dtype, B, N, L, H):
    from instantrestore_amd import ops
    from oracle import shared_attn_oracle as O
    C = H * 64
    g = torch.Generator().manual_seed(7 * L + N)
    ws = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dtype).cuda()
    wc = (torch.randn(3 * C, C, generator=g) / C ** 0.5 * 1.7).to(dtype).cuda()
    hs = torch.randn(B, L, C, generator=g).cuda()
    hr = (torch.randn(B * N, L, C, generator=g) + 0.1).cuda()
    ys, st_s = ops.linear(hs, ws, None, stats=(2 * C, C))
    yr, st_c = ops.linear(hr, wc, None, stats=(2 * C, C))
    v, rv = ys[..., 2 * C:], yr[..., 2 * C:].reshape(B, N, L, C)
    valid = torch.tensor([N if b % 2 == 0 else max(1, N - 2) for b in range(B)], dtype=torch.int32)
    # reference: float64 oracle on the rounded tensors, invalid references zero-filled like pix2pix_turbo.py:269-273
    rv_z = rv.clone()
    for b in range(B):
        rv_z[b, int(valid[b]):] = 0
    f = lambda t: t.float().cpu().numpy().astype(np.float64)
    a_ref, b_ref = O.adain_affine_np(f(v), f(rv_z), H)
    a_ref, b_ref = a_ref.reshape(B, N, H, 64), b_ref.reshape(B, N, H, 64)
    # (1) content as partials + the valid counts
    a1, b1 = ops.adain_affine_from_partials(st_s, B, L, N, L, content=st_c, valid=valid.cuda())
    # (2) content as finished statistics, zeroed by the harvest for the invalid references
    cm, cs = ops.token_stats_from_partials(st_c, B * N, L)
    keep = (torch.arange(N)[None, :] < valid[:, None]).float().cuda()[:, :, None, None]
    cm, cs = cm.reshape(B, N, H, 64) * keep, cs.reshape(B, N, H, 64) * keep
    a2, b2 = ops.adain_affine_from_partials(st_s, B, L, N, L, content_mean=cm.contiguous(), content_std=cs.contiguous())
    # (3) the standalone pass of rounds 1-3 on the zero-filled tensor
    a0, b0 = ops.adain_stats(v, rv_z, heads=H)
    for a, b_, tag in ((a1, b1, "partials"), (a2, b2, "finished statistics"), (a0, b0, "standalone")):
        an, bn = a.cpu().numpy().astype(np.float64), b_.cpu().numpy().astype(np.float64)
        ok = np.isfinite(a_ref)
        assert np.abs(an - a_ref)[ok].max() <= 1e-5 * np.abs(a_ref)[ok].max(), tag       # 1e-5 relative
        assert np.abs(bn - b_ref)[ok].max() <= 1e-5 * max(1.0, np.abs(b_ref)[ok].max()), tag
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    # zero-filled reference: b == mean(V_self) exactly (attn_processors.py:242-246 on an all-zero V)
    ms, _ = ops.token_stats_from_partials(st_s, B, L)
    for b in range(B):
        for n in range(int(valid[b]), N):
            assert torch.equal(b1[b, n], ms[b])


def test_the_step_launches_no_statistics_pass_and_keeps_its_outputs():
    """bench step, cfg 2: with the fused tail no adain_partial / token-stats pass over V runs (the ops the processors
    call are counted), and every output stays within the 16-bit output rounding of the unfused path (the affine differs
    by fp32 merge order only where the GEMM's row block is not 256)"""
    import bench
    from instantrestore_amd import attn_processors as ap
    from instantrestore_amd import ops
    dev = torch.device("cuda", 0)
    layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=5)
    saved = bench._AUTOCAST["dtype"]
    bench._AUTOCAST["dtype"] = dtype
    calls = {"adain_stats": 0, "adain_stats_cached": 0, "token_stats": 0, "affine": 0, "tsp": 0}
    orig = (ops.adain_stats, ops.adain_stats_cached, ops.token_stats, ops.adain_affine_from_partials, ops.token_stats_from_partials)

    def count(name, fn):
        def w(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return w
    try:
        with torch.no_grad():
            ap.FUSED_STATS = False
            ref = [o.clone() for o in bench.hot_path_step(layers, B, N, False, False)]
            ap.FUSED_STATS = True
            ops.adain_stats, ops.adain_stats_cached, ops.token_stats = count("adain_stats", orig[0]), count("adain_stats_cached", orig[1]), count("token_stats", orig[2])
            ops.adain_affine_from_partials, ops.token_stats_from_partials = count("affine", orig[3]), count("tsp", orig[4])
            for two in (False, True):
                got = bench.hot_path_step(layers, B, N, False, two)
                torch.cuda.synchronize()
                for a, b in zip(ref, got):
                    assert (a.float() - b.float()).abs().max() <= 2.0 ** -7 * max(1.0, float(a.float().abs().max()))
    finally:
        ap.FUSED_STATS = True
        ops.adain_stats, ops.adain_stats_cached, ops.token_stats, ops.adain_affine_from_partials, ops.token_stats_from_partials = orig
        bench._AUTOCAST["dtype"] = saved
    assert calls["adain_stats"] == calls["adain_stats_cached"] == calls["token_stats"] == 0, calls
    # one launch per shared layer merges both sides' partials into the affine; nothing is launched on the capture side
    assert calls["affine"] == 18 and calls["tsp"] == 0, calls


@pytest.mark.parametrize("C_", [320, 640], ids=["K320", "K640"])
def test_row_count_that_leaves_waves_without_a_block_writes_no_partial_out_of_range(C_):
    """M = 65600 = 1025 blocks of 64 rows: whole statistics blocks, but not whole 256-row workgroups of the X-stationary kernels -
    the last workgroup has three waves past M.  They own no block and no slot in the workspace: the canary behind the
    workspace must survive, and the partials must still merge to the statistics of the rounded V."""
    import ctypes as C
    from instantrestore_amd import _lib, ops
    H = C_ // 64
    sets, L = 1025, 64
    M = sets * L
    g = torch.Generator().manual_seed(99 + C_)
    w = (torch.randn(3 * C_, C_, generator=g) / C_ ** 0.5).to(torch.bfloat16).cuda()
    x = torch.randn(M, C_, generator=g).cuda()
    assert ops.linear_kernel_for(M, 3 * C_, C_, False) == 1 and ops.linear_stats_rows(M, 3 * C_, C_, False) == 64   # X-stationary
    need = (M // 64) * H * 128
    ws = torch.full((need + 8192,), 12345.0, dtype=torch.float32, device="cuda")
    y = torch.empty(M, 3 * C_, dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().ir_linear_fwd_stats(1, 1, M, 3 * C_, C_, x.data_ptr(), C_, w.data_ptr(), C_, None, y.data_ptr(), 3 * C_, 0, 1.0,
                                        2 * C_, C_, ws.data_ptr(), need * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, _lib.lib().ir_last_error_string()
    torch.cuda.synchronize()
    assert bool((ws[need:] == 12345.0).all()), "a partial was written behind the workspace"
    st = ops.ColumnStats(ws[:need].view(M // 64, H, 128), 64, H)
    m1, s1 = ops.token_stats_from_partials(st, sets, L)
    v = y[:, 2 * C_:].reshape(sets, L, H, 64).float()
    assert torch.equal(y, ops.linear(x, w, None))
    assert _rel(m1.cpu(), v.mean(1).cpu()) <= 1e-5 and _rel(s1.cpu(), v.std(1, unbiased=True).cpu()) <= 1e-5
