#!/usr/bin/env python3
"""Where the HOST time of an eager step goes (one identity, fp16 autocast: the way inference/test.py:79-111 drives the model - the
launch-bound case): cProfile over N eager one-stream steps of bench.py's cfg1gpu workload, GPU idle-waited only at the end.
usage: python tools/gpu_host_overhead.py [steps=200] [top=30]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg1gpu", True, dev, 1234)
bench._AUTOCAST["dtype"] = dtype
with torch.no_grad():
    for _ in range(5):
        bench.hot_path_step(layers, B, N)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bench.hot_path_step(layers, B, N)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"# {steps} eager one-stream steps, B={B} N={N} {px}px {dtype}: host issue {1e3 * t_issue / steps:.3f} ms/step, wall {1e3 * t_all / steps:.3f} ms/step")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        bench.hot_path_step(layers, B, N)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(top)
print(s.getvalue()[:9000])
