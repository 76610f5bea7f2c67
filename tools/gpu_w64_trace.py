#!/usr/bin/env python3
"""Phase timeline of the default 64-row kernel (pre-scaled Q instantiation) from a -DW64_TRACE build (development
aid): s_memtime stamps of all 8 waves of workgroup 0 at the phase boundaries of tiles 8..23.
usage: IR_LIB_PATH=<trace build> gpu_w64_trace.py [adain]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from instantrestore_amd import ops
B, N, L, H = 8, 4, 4096, 5
C = H * 64
torch.manual_seed(0)
dt = torch.bfloat16
q, k, v = (torch.randn(B, L, C, device="cuda").to(dt) for _ in range(3))
rk = torch.randn(B, N, L, C, device="cuda").to(dt)
rv = torch.randn(B, N, L, C, device="cuda").to(dt)
adain = None
if "adain" in sys.argv:
    adain = ops.adain_stats(v, rv, heads=H) if hasattr(ops, "adain_stats") else None
qs = (q.float() * (0.125 * 1.4426950408889634)).to(dt)
for _ in range(3):
    out, lse = ops.shared_attention(qs, k, v, rk, rv, heads=H, scale=0.125, include_self=True, adain=adain,
                                    return_lse=True, q_prescaled=True)
torch.cuda.synchronize()
tr = lse.flatten()[:8 * 16 * 8].view(torch.int32).cpu().view(8, 16, 8).numpy().astype("int64") & 0xffffffff
ck = lse.flatten()[1024:1027].view(torch.int32).cpu().numpy().astype("int64") & 0xffffffff
print("workgroup 0: %d shader cycles in %d ticks of 100 MHz over %d tiles -> %.3f GHz, %.0f cycles per tile" % (ck[0], ck[1], ck[2], ck[0] / ck[1] * 0.1, ck[0] / max(ck[2], 1)))
names = ["issue", "QK", "softmax+PV", "fold/vmcnt", "barrier"]
tot = np.zeros(5)
cnt = 0
for w in range(8):
    print("wave", w)
    for t in range(1, 15):
        r = tr[w, t]
        nxt = tr[w, t + 1][0]
        d = [r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], nxt - r[4]]
        d = [int(x) % (1 << 32) for x in d]
        tot += np.array(d); cnt += 1
        print("  tile %2d: issue %5d  QK %5d  softmax+PV %5d  vmcnt %5d  barrier %5d | period %5d   start-w0 %6d" % (t + 8, *d, sum(d), int(r[0] - tr[0, t, 0])))
print("mean per tile:", {n: round(x / cnt) for n, x in zip(names, tot)}, "period", round(tot.sum() / cnt))
