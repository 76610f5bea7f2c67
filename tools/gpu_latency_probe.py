"""single-identity latency of the hot-path step (the reference's own inference case, test.py: one identity at a
time): eager one stream, eager two streams, and the two-stream step replayed from one hipGraph"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
bench.CONFIGS["lat"] = (B, 4, 512, "f16", True)
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("lat", True, dev, seed=1)

def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    print(f"B={B}: eager, one stream : {timed(lambda: bench.hot_path_step(layers, B, N, False, False)):.3f} ms/step")
    print(f"B={B}: eager, two streams: {timed(lambda: bench.hot_path_step(layers, B, N, False, True)):.3f} ms/step")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bench._REF_STREAM.clear()
        bench.hot_path_step(layers, B, N, False, True)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            bench.hot_path_step(layers, B, N, False, True)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    print(f"B={B}: hipGraph replay   : {timed(g.replay):.3f} ms/step")
