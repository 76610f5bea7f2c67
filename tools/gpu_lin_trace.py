#!/usr/bin/env python3
"""Phase timeline of the K <= 320 projection GEMM from a -DLIN_TRACE build (development aid): s_memtime stamps of the four
waves of workgroup 0 over chunks 4..19.  usage: IR_LIB_PATH=<trace build> gpu_lin_trace.py [M=131072] [N=960] [f32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from instantrestore_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 960
xdt = torch.float32 if "f32" in sys.argv else torch.bfloat16
x = torch.randn(M, 320, device="cuda", dtype=xdt)
w = torch.randn(N, 320, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    y = ops.linear(x, w)
torch.cuda.synchronize()
tr = y.flatten()[:4 * 16 * 8 * 2].view(torch.int32).cpu().view(4, 16, 8).numpy().astype("int64") & 0xffffffff
names = ["DMA issue", "staging", "MFMA+stores", "vmcnt", "barrier"]
tot = np.zeros(5); cnt = 0
for wv in range(4):
    print("wave", wv)
    for t in range(1, 15):
        r = tr[wv, t]; nxt = tr[wv, t + 1][0]
        d = [int(x) % (1 << 32) for x in (r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], nxt - r[4])]
        tot += np.array(d); cnt += 1
        print("  chunk %2d: issue %5d  staging %5d  MFMA+stores %5d  vmcnt %5d  barrier %5d | period %5d" % (t + 4, *d, sum(d)))
print("mean per chunk:", {n: round(v / cnt) for n, v in zip(names, tot)}, "period", round(tot.sum() / cnt))
