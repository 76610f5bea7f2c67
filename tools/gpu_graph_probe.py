"""experiment: the whole two-stream step captured in one hipGraph and replayed"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=1234)
with torch.no_grad():
    for _ in range(5):
        outs = bench.hot_path_step(layers, B, N, False, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        outs = bench.hot_path_step(layers, B, N, False, True)
    torch.cuda.synchronize()
    print("eager two streams: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
    ref = [o.clone() for o in outs]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        bench._REF_STREAM.clear()
        bench.hot_path_step(layers, B, N, False, True)   # side stream + workspaces for this capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            gouts = bench.hot_path_step(layers, B, N, False, True)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph replay     : %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
    print("same outputs:", all(torch.equal(a, b) for a, b in zip(ref, gouts)))
