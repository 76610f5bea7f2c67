"""interleaved, SUSTAINED A/B of GEMM kernels on one shape: each variant runs back to back for `secs` (the board settles at its
power cap within milliseconds; short bursts ride on the idle clock), `rounds` times in rotation.
usage: gpu_gemm_ab_sustained.py M N K bias(0/1) fp32(0/1) kernel [kernel ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
M, N, K, bias, f32 = (int(a) for a in sys.argv[1:6])
kernels = sys.argv[6:]
secs, rounds = float(os.environ.get("SECS", "1.0")), int(os.environ.get("ROUNDS", "3"))
x = torch.randn(M, K, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
b = torch.randn(N, device="cuda").to(torch.bfloat16) if bias else None
res = {k: [] for k in kernels}
for r in range(rounds):
    for k in kernels:
        kid = ops.LIN_KERNELS[k]
        for _ in range(20):
            ops.linear(x, w, b, kernel=kid)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(200):
                ops.linear(x, w, b, kernel=kid)
            n += 200
            torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / n * 1e6)
for k in kernels:
    v = sorted(res[k])
    print(f"{M}x{N}x{K} bias={bias} x_fp32={f32} {k:18s} median {v[len(v)//2]:7.1f} us  {['%.1f' % t for t in res[k]]}")
