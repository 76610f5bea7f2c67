"""AdaIN content statistics computed ONCE per identity: the K/V-capture layer emits (mean, std) of every
reference V (ir_token_stats), the shared layer reads only V_self (ir_adain_stats_cached).  The affine must be the SAME BITS as
ir_adain_stats on the same tensors (same partial kernel, same merge order), the zero-reference quirk of the reference
(attn_processors.py:242-246 on a zero-filled V: b == mean(V_self) exactly, pix2pix_turbo.py:269-273) must survive the cache -
the zero fill invalidates the cached statistics of the zeroed references - and the shared processor's output must not change
by a bit when ``ref_stats`` is handed over."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("B,H,L,N,Lr", [(8, 20, 256, 4, 256), (8, 10, 1024, 4, 1024), (8, 5, 4096, 4, 4096), (2, 3, 300, 3, 300),
                                        (1, 2, 200, 8, 200), (2, 5, 16384, 2, 16384), (3, 2, 7, 2, 7)])
def test_cached_affine_is_bit_identical(dtype, B, H, L, N, Lr):
    from instantrestore_amd import ops
    g = torch.Generator().manual_seed(L + N)
    C = H * 64
    v = (torch.randn(B, L, C, generator=g) * 0.8 + 0.3).to(dtype).cuda()
    rv = (torch.randn(B, N, Lr, C, generator=g) * 1.7 - 0.2).to(dtype).cuda()
    a0, b0 = ops.adain_stats(v, rv, heads=H)
    # statistics as the capture layer stashes them: one (B*N, 1, L, C) call over the flat reference token sets
    m, sd = ops.token_stats(rv.reshape(B * N, 1, Lr, C), heads=H)
    a1, b1 = ops.adain_stats_cached(v, m.reshape(B, N, H, 64), sd.reshape(B, N, H, 64), heads=H)
    assert torch.equal(a0, a1) and torch.equal(b0, b1)
    # the K/V stash is a strided view of the fused q/k/v output: same bits from the view
    qkv = torch.randn(B * N, Lr, 3 * C, generator=g).to(dtype).cuda()
    m2, sd2 = ops.token_stats(qkv[..., 2 * C:].unsqueeze(1), heads=H)
    m3, sd3 = ops.token_stats(qkv[..., 2 * C:].contiguous().unsqueeze(1), heads=H)
    assert torch.equal(m2, m3) and torch.equal(sd2, sd3)


@pytest.mark.parametrize("fused", [False, True], ids=["standalone_stats", "gemm_tail_stats"])
def test_zero_filled_reference_keeps_the_style_mean_quirk_through_the_cache(fused):
    """``fused``: the capture layers' content statistics come from their q/k/v GEMM's tail (round 4) instead of the
    standalone pass - same values to fp32 rounding (64-row blocks, another merge order), not the same bits"""
    from face_replace.models.attn_processors import register_attention_processor_kv_unet
    from instantrestore_amd import attn_processors as ap
    from instantrestore_amd import ops
    from instantrestore_amd.kv_harvest import get_conditioning_keys_values
    from instantrestore_amd.unet_host import AttnTopologyUNet
    import __graft_entry__ as ge
    from types import SimpleNamespace
    cfg = SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False)
    unet = AttnTopologyUNet(seed=3).to("cuda")
    ge.register_attention_processor_kv_unet_default(unet, cfg)
    register_attention_processor_kv_unet(unet)
    text = torch.randn(4, 77, 1024, device="cuda")
    ap.FUSED_STATS = fused
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            keys, vals, stats = get_conditioning_keys_values(unet, torch.randn(4, 4, 16, 16, device="cuda"), None, text, 2, [2, 1],
                                                             with_stats=True)
    finally:
        ap.FUSED_STATS = True
    assert len(stats) == 9
    for l, (v, st) in enumerate(zip(vals, stats)):
        B, N, L, C = v.shape
        # GEMM partials travel as they are (.finished() = the (mean, std) pair) wherever the token axis is whole 64-row blocks;
        # the 4 x 4-token layers of this 16 x 16 latent keep the standalone pass
        assert hasattr(st, "finished") == bool(fused and L % 64 == 0)
        st = st.finished() if hasattr(st, "finished") else st
        H = C // 64
        assert st[0].shape == (B, N, H, 64) and float(st[0][1, 1].abs().max()) == 0.0 and float(st[1][1, 1].abs().max()) == 0.0
        vs = torch.randn(B, L, C, device="cuda").to(v.dtype)
        a0, b0 = ops.adain_stats(vs, v, heads=H)                       # from the zero-filled tensor itself
        a1, b1 = ops.adain_stats_cached(vs, st[0], st[1], heads=H)     # from the (invalidated) cached statistics
        if not fused:
            assert torch.equal(a0, a1) and torch.equal(b0, b1)
        else:    # 1e-5 relative (floating point: fp32 statistics merged in another order)
            fin = torch.isfinite(a0)
            assert float((a0 - a1)[fin].abs().max()) <= 1e-5 * float(a0[fin].abs().max())
            assert float((b0 - b1)[fin].abs().max()) <= 1e-5 * max(1.0, float(b0[fin].abs().max()))
        mean_s, _ = ops.token_stats(vs.unsqueeze(1), heads=H)
        assert torch.equal(b1[1, 1], mean_s[1, 0])                     # zeroed reference: b == mean(V_self) exactly


def test_shared_processor_output_is_unchanged_by_ref_stats(two_streams=False):
    import bench
    from instantrestore_amd import attn_processors as ap
    dev = torch.device("cuda", 0)
    layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=77)
    saved = bench._AUTOCAST["dtype"]
    bench._AUTOCAST["dtype"] = dtype
    ap.FUSED_STATS = False     # the round-3 path (standalone statistics passes): its cached form is bit-identical; the
                               # round-4 GEMM-tail form is held to its own tests (tests/test_gpu_fused_stats.py)
    try:
        with torch.no_grad():
            bench.REF_STATS["on"] = False
            ref = [o.clone() for o in bench.hot_path_step(layers, B, N, False, two_streams)]
            bench.REF_STATS["on"] = True
            got = bench.hot_path_step(layers, B, N, False, two_streams)
            torch.cuda.synchronize()
    finally:
        ap.FUSED_STATS = True
        bench.REF_STATS["on"] = True
        bench._AUTOCAST["dtype"] = saved
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_randomised_statistics_shapes_against_float64():
    """seeded sweep (IR_SWEEP_CASES / IR_SWEEP_SEED widen it) over the standalone statistics kernels: token axes from 2 tokens to
    several 256-row chunks and ragged ends, different lengths for style and content, channel means far from zero, strided views,
    a zero-filled reference now and then.  ``ir_token_stats`` against float64 mean / unbiased std (1e-5 relative to the largest
    entry), ``ir_adain_stats`` against the oracle's affine applied to the content (the reference's ``adain`` on float64,
    attn_processors.py:7-18: 2e-4 relative), the cached form bit-identical to it, ``ir_adain_apply`` against x * a + b."""
    import os
    import numpy as np
    from instantrestore_amd import ops
    from oracle import shared_attn_oracle as O
    seed = int(os.environ.get("IR_SWEEP_SEED", "808"))
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    for case in range(int(os.environ.get("IR_SWEEP_CASES", "40"))):
        dtype = [torch.bfloat16, torch.float16][case % 2]
        B, N, H = int(rng.integers(1, 4)), int(rng.integers(1, 6)), int(rng.integers(1, 5))
        pick = lambda: int(rng.choice([rng.integers(8, 64), rng.integers(64, 600), 256 * rng.integers(1, 5) + rng.integers(-1, 2), rng.integers(600, 3000)]))
        Ls, Lr = pick(), pick()
        C = H * 64
        shift_v, shift_x = float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3))
        sv, sx = float(rng.uniform(0.2, 2.0)), float(rng.uniform(0.2, 2.0))
        v = (torch.randn(B, Ls, C, generator=g) * sv + shift_v).to(dtype)
        rv = (torch.randn(B, N, Lr, C, generator=g) * sx + shift_x).to(dtype)
        if rng.integers(0, 4) == 0:
            rv[int(rng.integers(0, B)), int(rng.integers(0, N))] = 0          # zero-filled reference: (a, b) -> (~0, mean(V_self))
        what = f"stats case {case}: B{B} N{N} H{H} Ls{Ls} Lr{Lr} shifts {shift_v:.2f}/{shift_x:.2f} {dtype}"
        if rng.integers(0, 2):    # strided views (the V third of a fused projection output)
            bv = torch.zeros(B, Ls, 3 * C, dtype=dtype, device="cuda"); bv[..., 2 * C:] = v.cuda(); vd = bv[..., 2 * C:]
            br = torch.zeros(B, N, Lr, 3 * C, dtype=dtype, device="cuda"); br[..., 2 * C:] = rv.cuda(); rvd = br[..., 2 * C:]
        else:
            vd, rvd = v.cuda(), rv.cuda()
        v64, rv64 = v.double().numpy(), rv.double().numpy()
        m, sd = ops.token_stats(rvd, heads=H)
        m_ref, sd_ref = rv64.mean(2), rv64.std(2, ddof=1)
        assert np.abs(m.cpu().numpy().reshape(B, N, C) - m_ref).max() <= 1e-5 * max(1.0, np.abs(m_ref).max()), what
        assert np.abs(sd.cpu().numpy().reshape(B, N, C) - sd_ref).max() <= 1e-5 * max(1.0, np.abs(sd_ref).max()) + 2e-6 * abs(shift_x), what
        a, b = ops.adain_stats(vd, rvd, heads=H)
        a1, b1 = ops.adain_stats_cached(vd, m, sd, heads=H)
        assert torch.equal(a, a1) and torch.equal(b, b1), what
        a_ref, b_ref = O.adain_affine_np(v64, rv64, H)
        y_ref = rv64 * a_ref[:, :, None, :] + b_ref[:, :, None, :]
        an, bn = a.cpu().numpy().astype(np.float64).reshape(B, N, 1, C), b.cpu().numpy().astype(np.float64).reshape(B, N, 1, C)
        y_fp = rv64 * an + bn                                                   # the device's affine on the same content
        assert np.abs(y_fp - y_ref).max() <= 2e-4 * max(1.0, np.abs(y_ref).max()), (what, np.abs(y_fp - y_ref).max())
        y = ops.adain_apply(rvd, a, b, heads=H)
        tol = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7) * max(1.0, np.abs(y_ref).max())
        assert np.abs(y.double().cpu().numpy() - y_fp).max() <= tol, what
