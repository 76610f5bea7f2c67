"""HBM streaming rates seen by plain kernels on this box (context for the HBM-bound rooflines)."""
import torch
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (256, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, dtype=torch.bfloat16, device="cuda"); b = torch.empty_like(a)
    a.normal_()
    t_fill = timeit(lambda: b.fill_(1.0))
    t_copy = timeit(lambda: b.copy_(a))
    t_read = timeit(lambda: a.sum())
    print(f"{mb:5d} MiB: fill {mb/1024*1.0737/t_fill*1e3:6.2f} TB/s write | copy {2*mb/1024*1.0737/t_copy*1e3:6.2f} TB/s (r+w) | sum {mb/1024*1.0737/t_read*1e3:6.2f} TB/s read")
