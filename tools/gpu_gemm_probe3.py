"""Round 3: every projection GEMM shape of the cfg-2 step (and cfg 5's large ones with `big`) through every kernel of
ir_linear_fwd_ex - X-stationary, the five tile shapes of the LDS-tiled kernel - with 16-bit and fp32 activations, next to
the vendor GEMM (F.linear, bf16 x; + the separate cast pass it needs for fp32 x).  HIP-event timing, median of 5 x 20
launches; every own result is also compared with the vendor's.  usage: python tools/gpu_gemm_probe3.py [big]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from instantrestore_amd import ops


def timeit(fn, iters=20, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if os.environ.get("GRAPH") == "1":      # round 6: the launches replayed from one hipGraph - device time of small GEMMs without
        g_ = torch.cuda.CUDAGraph()         # the ~12-us eager launch cadence on top (what a captured B = 1 step sees)
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g_, stream=s_):
                for _ in range(iters):
                    fn()
        torch.cuda.current_stream().wait_stream(s_)
        g_.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g_.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / iters)
        return sorted(ts)[len(ts) // 2] * 1e3
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2] * 1e3   # us


shapes = []
for (L, C) in ((256, 1280), (1024, 640), (4096, 320)):
    for sets in (32, 8):
        shapes += [(sets * L, 3 * C, C, False), (sets * L, C, C, True)]
if len(sys.argv) > 1 and sys.argv[1] == "big":
    shapes = [(128 * 1024, 3840, 1280, False), (128 * 1024, 1280, 1280, True), (16 * 1024, 3840, 1280, False)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    # the tiny-M regime the round-3 probes never measured - cross-attention to_k / to_v over the text states
    # (M = 77 B, K = 1024), the mid block's projections (8 x 8 tokens: M = 64 B), one identity (M = L)
    shapes = [(M, N, K, b) for K in (1024, 1280) for M in (77, 616, 1024) for (N, b) in ((1280, False), (3840, False), (1280, True))
              if not (K == 1024 and N == 3840)]
    shapes += [(512, 3840, 1280, False), (512, 1280, 1280, True), (256, 960, 320, False), (4096, 960, 320, False), (64, 3840, 1280, False)]
if len(sys.argv) > 1 and sys.argv[1] == "b1":
    # round 6: ONE identity with 4 references (cfg1gpu): capture layers M = 4 L, shared layers M = L
    shapes = []
    for (L, C) in ((256, 1280), (1024, 640), (4096, 320)):
        for sets in (4, 1):
            shapes += [(sets * L, 3 * C, C, False), (sets * L, C, C, True)]
dt = torch.bfloat16
g = torch.Generator().manual_seed(1)
tot_v = tot_o = 0.0
for (M, N, K, bias) in shapes:
    x32 = torch.randn(M, K, generator=g).cuda()
    x = x32.to(dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(dt)
    b = torch.randn(N, generator=g).cuda().to(dt) if bias else None
    ref = F.linear(x, w, b)
    t_v = timeit(lambda: F.linear(x, w, b))
    t_c = timeit(lambda: x32.to(dt))
    fl = 2.0 * M * N * K
    row = [f"M={M:6d} N={N:5d} K={K:5d} bias={int(bias)} | vendor {t_v:6.1f} us ({fl / t_v / 1e6:6.0f} TF/s) + cast {t_c:5.1f} |"]
    auto = ops.linear_kernel_for(M, N, K, bias)
    best = None
    for name, kid in ops.LIN_KERNELS.items():
        if kid == 0:
            continue
        try:
            y = ops.linear(x, w, b, kernel=kid)
        except Exception:
            continue
        err = float((y.float() - ref.float()).abs().max())
        y32 = ops.linear(x32, w, b, kernel=kid)
        same32 = bool(torch.equal(y32, y))
        t16 = timeit(lambda: ops.linear(x, w, b, kernel=kid))
        t32 = timeit(lambda: ops.linear(x32, w, b, kernel=kid))
        row.append(f" {name}{'*' if kid == auto else ''}: {t16:6.1f}/{t32:6.1f} ({fl / t16 / 1e6:5.0f} TF/s) d={err:.3g}{'' if same32 else ' F32!='}")
        if best is None or t32 < best[1]:
            best = (name, t32, t16)
    print("".join(row), f" || best(fp32 x) {best[0]} {best[1]:.1f}", flush=True)
    tot_v += 3 * (t_v + (0.0 if bias else t_c))     # q/k/v inputs are fp32 under autocast (cast pass), out inputs 16 bit
    tot_o += 3 * (best[2] if bias else best[1])
print(f"per step (x3 layers): vendor + casts {tot_v / 1e3:.3f} ms, best own kernel per shape {tot_o / 1e3:.3f} ms")
