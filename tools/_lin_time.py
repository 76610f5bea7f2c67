import sys, os
sys.path.insert(0, "/root/repo")
import torch
from instantrestore_amd import ops
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tag = os.environ.get("IR_LIB_PATH", "default")[-20:]
for (M, K, N) in ((131072, 320, 960), (131072, 320, 320), (32768, 320, 960), (32768, 640, 1920), (8192, 640, 1920)):
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for xdt in (torch.float32, torch.bfloat16):
        x = torch.randn(M, K, device="cuda", dtype=xdt)
        ms = min(timeit(lambda: ops.linear(x, w)) for _ in range(3))
        by = M * K * x.element_size() + 2 * M * N
        print(f"{tag} M={M} K={K} N={N} x={str(xdt)[6:]}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.0f} TF/s  {by/ms/1e6:6.0f} GB/s")
