"""How close hipBLASLt gets to the roofline on the projection GEMMs of the step (bf16)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
import torch.nn.functional as F

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

tot = 0.0; ideal_tot = 0.0
for (L, C) in ((4096, 320), (1024, 640), (256, 1280)):
    for sets, tag in ((32, "refs"), (8, "ident")):
        M = sets * L
        x = torch.randn(M, C, device="cuda", dtype=torch.bfloat16)
        for N, bias, name in ((3 * C, False, "qkv"), (C, True, "out")):
            w = torch.randn(N, C, device="cuda", dtype=torch.bfloat16)
            b = torch.randn(N, device="cuda", dtype=torch.bfloat16) if bias else None
            ms = timeit(lambda: F.linear(x, w, b))
            fl = 2.0 * M * N * C; by = 2.0 * (M * C + N * C + M * N)
            ideal = max(fl / 2.5e15, by / 6.3e12) * 1e3
            tot += 3 * ms; ideal_tot += 3 * ideal
            print(f"{name:4s} {tag:5s} M={M:6d} N={N:5d} K={C:5d}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF/s  {by/ms/1e6:7.1f} GB/s  roofline {ideal*1e3:6.1f} us ({ideal/ms*100:4.1f}%)")
print(f"per step (x3 layers each): {tot:.3f} ms measured, {ideal_tot:.3f} ms at the roofline")

# the same K = 320 shapes through ir_linear_fwd
from instantrestore_amd import ops
for sets, tag in ((32, "refs"), (8, "ident"), (128, "refs 1024px")):
    M = sets * 4096
    x = torch.randn(M, 320, device="cuda", dtype=torch.bfloat16)
    for N, bias, name in ((960, False, "qkv"), (320, True, "out")):
        w = torch.randn(N, 320, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(N, device="cuda", dtype=torch.bfloat16) if bias else None
        ms = timeit(lambda: ops.linear(x, w, b))
        ms_lib = timeit(lambda: F.linear(x, w, b))
        fl = 2.0 * M * N * 320; by = 2.0 * (M * 320 + N * 320 + M * N)
        ideal = max(fl / 2.5e15, by / 6.3e12) * 1e3
        print(f"ir_linear {name:4s} {tag:11s} M={M:6d} N={N:4d}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF/s {by/ms/1e6:7.1f} GB/s ({ideal/ms*100:4.1f}% of roofline) | vendor {ms_lib*1e3:7.1f} us")

# the K = 640 shapes through ir_linear_fwd's split-contraction form
for sets, tag in ((32, "refs"), (8, "ident"), (128, "refs 1024px")):
    M = sets * 1024
    x = torch.randn(M, 640, device="cuda", dtype=torch.bfloat16)
    for N, bias, name in ((1920, False, "qkv"), (640, True, "out")):
        w = torch.randn(N, 640, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(N, device="cuda", dtype=torch.bfloat16) if bias else None
        ms = timeit(lambda: ops.linear(x, w, b))
        ms_lib = timeit(lambda: F.linear(x, w, b))
        fl = 2.0 * M * N * 640
        print(f"ir_linear K=640 {name:4s} {tag:11s} M={M:6d} N={N:4d}: {ms*1e3:7.1f} us  {fl/ms/1e9:7.1f} TF/s | vendor {ms_lib*1e3:7.1f} us")

