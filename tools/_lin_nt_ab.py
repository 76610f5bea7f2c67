"""A/B of the GEMM output stores' cache policy: runs the step-shape list through ops.linear (automatic kernel), fp32 and 16-bit x.
usage: [IR_LIB_PATH=...] python tools/_lin_nt_ab.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from instantrestore_amd import ops
def timeit(fn, iters=20, reps=5):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / iters)
    return sorted(ts)[len(ts) // 2] * 1e3
shapes = []
for (L, C) in ((256, 1280), (1024, 640), (4096, 320)):
    for sets in (32, 8):
        shapes += [(sets * L, 3 * C, C, False), (sets * L, C, C, True)]
g = torch.Generator().manual_seed(1); dt = torch.bfloat16; tot = 0.0; out = []
for (M, N, K, bias) in shapes:
    x32 = torch.randn(M, K, generator=g).cuda(); x = x32.to(dt)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(dt)
    b = torch.randn(N, generator=g).cuda().to(dt) if bias else None
    t16, t32 = timeit(lambda: ops.linear(x, w, b)), timeit(lambda: ops.linear(x32, w, b))
    tot += 3 * (t16 if bias else t32)
    out.append(f"{M}x{N}x{K}:{t16:.1f}/{t32:.1f}")
print(os.environ.get("IR_LIB_PATH", "product")[-20:], " ".join(out), f"| step total {tot / 1e3:.3f} ms", flush=True)
