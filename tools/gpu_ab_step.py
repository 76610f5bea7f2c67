"""same-box interleaved A/B of the bench step: usage gpu_ab_step.py <switch> ; switch in {ref_stats, own_gemm, fused_stats, x_stationary_rot, x_stationary_pp}
 fused_stats: AdaIN statistics as the tail of the q/k/v GEMMs (round 4) vs the standalone passes of round 3 (GRAPH=1: hipGraph replays)
 ref_stats: AdaIN content statistics from the capture layer (round 3) vs re-read in every shared layer
 own_gemm : this library's GEMMs for every projection vs F.linear for the shapes the vendor GEMM served before round 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from instantrestore_amd import attn_processors as ap, ops

what = sys.argv[1] if len(sys.argv) > 1 else "ref_stats"
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=1234)
bench._AUTOCAST["dtype"] = dtype
orig_supported = ops.linear_supported


def old_rule(x, w, b):   # rounds 1-2: own kernels for K <= 320 / 640 above 2^24 output elements only
    rows = x.numel() // x.shape[-1]
    k = w.shape[1]
    return orig_supported(x, w, b) and ((k % 64 == 0 and k <= 320) or k == 640) and rows * w.shape[0] >= (1 << 24)


orig_linear = ops.linear
XS_VARIANT = ops.LIN_KERNELS.get(what, 0)


def variant_linear(x, w, b=None, **kw):   # the K = 320 shapes the automatic choice gives to the X-stationary kernel -> the variant under test
    if kw.get("kernel", 0) == 0 and w.shape[1] == 320 and ops.linear_kernel_for(x.numel() // 320, w.shape[0], 320, b is not None) == 1:
        kw["kernel"] = XS_VARIANT
    return orig_linear(x, w, b, **kw)


def setmode(on):
    if what == "ref_stats":
        bench.REF_STATS["on"] = on
    elif what == "fused_stats":
        ap.FUSED_STATS = on
    elif XS_VARIANT:
        ops.linear = variant_linear if on else orig_linear
    else:
        ops.linear_supported = orig_supported if on else old_rule


def run(two, steps=20):
    if os.environ.get("GRAPH") == "1":   # one hipGraph of the step per mode, replayed
        with torch.no_grad():
            cap = bench.CapturedStep(layers, B, N, two)
            for _ in range(3):
                cap.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                cap.replay()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    with torch.no_grad():
        for _ in range(3):
            bench.hot_path_step(layers, B, N, False, two)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            bench.hot_path_step(layers, B, N, False, two)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {}
for rnd in range(4):
    for on in (True, False):
        setmode(on)
        for two in (True, False):
            res.setdefault((on, two), []).append(run(two))
setmode(True)
for (on, two), v in sorted(res.items()):
    print("%s=%-5s %s: %s ms  median %.3f" % (what, on, "two streams" if two else "one stream ", ["%.3f" % x for x in v], sorted(v)[len(v) // 2]))
