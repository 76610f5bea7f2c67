"""Is the bench step the same bits on one stream, on two streams and replayed from one hipGraph?  
Part 1: bench.determinism_report on cfg2.  Part 2, the bisect: every GEMM shape of the step that is NOT this library's
own kernel, run repeatedly through F.linear - alone, beside a busy second stream, and captured in a graph - and compared
bit for bit; then the same for the own kernels.  usage: python tools/gpu_determinism.py [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
layers, (B, N, px, dtype, use_adain) = bench.build_workload("cfg2", True, dev, seed=1234)
bench._AUTOCAST["dtype"] = dtype
with torch.no_grad():
    for _ in range(3):
        bench.hot_path_step(layers, B, N, False, True)
    torch.cuda.synchronize()
    rep = bench.determinism_report(layers, B, N, reps)
print("STEP", json.dumps(rep))

# ---- part 2: GEMM shapes --------------------------------------------------------------------------------------------
from instantrestore_amd import ops

shapes = [(8192, 3840, 1280, False), (8192, 1280, 1280, True), (2048, 3840, 1280, False), (2048, 1280, 1280, True),
          (8192, 1920, 640, False), (8192, 640, 640, True), (32768, 320, 320, True),
          (32768, 960, 320, False), (131072, 960, 320, False), (32768, 1920, 640, False)]
g = torch.Generator().manual_seed(5)
side = torch.cuda.Stream()
big = torch.randn(8192, 8192, device=dev, dtype=dtype)


def busy():          # something large on the other stream while the GEMM under test runs
    with torch.cuda.stream(side):
        for _ in range(2):
            torch.mm(big, big)


for (M, Nn, K, has_bias) in shapes:
    x = torch.randn(M, K, generator=g).to(dev, dtype)
    w = (torch.randn(Nn, K, generator=g) / K ** 0.5).to(dev, dtype)
    b = torch.randn(Nn, generator=g).to(dev, dtype) if has_bias else None
    for name, fn in (("F.linear", lambda: F.linear(x, w, b)),
                     ("ops.linear", (lambda: ops.linear(x, w, b)) if ops.linear_supported(x, w, b) else None)):
        if fn is None:
            continue
        with torch.no_grad():
            base = fn().clone()
            torch.cuda.synchronize()
            alone = sum(0 if torch.equal(fn(), base) else 1 for _ in range(reps))
            torch.cuda.synchronize()
            beside = 0
            for _ in range(reps):
                busy()
                beside += 0 if torch.equal(fn(), base) else 1
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                fn()
                torch.cuda.synchronize()
                with torch.cuda.graph(gr, stream=s):
                    y = fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            graph = 0
            for _ in range(reps):
                busy()
                gr.replay()
                torch.cuda.synchronize()
                graph += 0 if torch.equal(y, base) else 1
            del gr
        print("GEMM %-10s M=%6d N=%5d K=%5d bias=%d : differs alone %d/%d, beside a busy stream %d/%d, graph replay %d/%d"
              % (name, M, Nn, K, has_bias, alone, reps, beside, reps, graph, reps), flush=True)
