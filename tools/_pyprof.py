import sys, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=1234)
bench._AUTOCAST["dtype"] = dtype
with torch.no_grad():
    for _ in range(5):
        bench.hot_path_step(layers, B, N, False, True)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        bench.hot_path_step(layers, B, N, False, True)
    pr.disable()
    torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
