"""HBM write efficiency by granularity: the projection GEMMs write 128-B (X-stationary: 64 columns of a row) or 512-B
(256-column tile) pieces at the output row stride.  Here: copy a (M, N) bf16 matrix (a) whole, (b) in column groups of
64 / 256 / N/3 columns, one group per launch (each launch touches every row once, `width` bytes per row).
usage: python tools/gpu_write_pattern_probe.py [M] [N]"""
import sys
import torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 960
src = torch.randn(M, N, device="cuda").to(torch.bfloat16)
dst = torch.empty_like(src)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


nbytes = M * N * 2
t = timeit(lambda: dst.copy_(src))
print(f"whole matrix copy ({nbytes / 1e6:.0f} MB read + write): {t:7.1f} us  {2 * nbytes / t / 1e6:6.2f} TB/s (read+write)")
t = timeit(lambda: dst.fill_(1.0))
print(f"whole matrix fill (write only)                    : {t:7.1f} us  {nbytes / t / 1e6:6.2f} TB/s")
for w in (64, 128, 256, N // 3):
    if N % w:
        continue
    groups = N // w
    def run():
        for j in range(groups):
            dst[:, j * w:(j + 1) * w].fill_(1.0)
    t = timeit(run, 5)
    print(f"fill in {groups:3d} launches of {w * 2:4d}-B pieces per row           : {t:7.1f} us  {nbytes / t / 1e6:6.2f} TB/s (incl. {groups} launch boundaries)")
