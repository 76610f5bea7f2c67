#!/usr/bin/env python3
"""Phase timeline of the ping-pong attention kernel (variant 15) from a -DW64_PP_TRACE build (development aid):
s_memtime stamps of waves 0 and 4 of workgroup 0 at the phase boundaries of tiles 8..23.
usage: IR_LIB_PATH=<trace build> gpu_pp_trace.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
B, N, L, H = 8, 4, 4096, 5
C = H * 64
torch.manual_seed(0)
dt = torch.bfloat16
q, k, v = (torch.randn(B, L, C, device="cuda").to(dt) for _ in range(3))
rk = torch.randn(B, N, L, C, device="cuda").to(dt)
rv = torch.randn(B, N, L, C, device="cuda").to(dt)
ops.set_attn_variant(15)
for _ in range(3):
    out, lse = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=True, adain=None, return_lse=True)
torch.cuda.synchronize()
tr = lse.flatten()[:2 * 16 * 8].view(torch.int32).cpu().view(2, 16, 8).numpy().astype("int64")
names = ["M start", "PV done", "QK done+vmcnt", "after barrier (V start)", "V done", "after barrier"]
for g in range(2):
    print("wave", 4 * g)
    for t in range(1, 15):
        r = tr[g, t]
        nxt = tr[g, t + 1][0]
        d = [(r[1] - r[0]), (r[2] - r[1]), (r[3] - r[2]), (r[4] - r[3]), (r[5] - r[4]), (nxt - r[0])]
        print("  tile %2d: PV %5d  QK %5d  barrier wait %5d  softmax %5d  barrier wait %5d | period %5d" % (t + 8, *[int(x) & 0xffffffff if x < 0 else int(x) for x in d]))
# relative offset of the two waves
print("wave4.Mstart - wave0.Mstart at tile 12:", int(tr[1, 4, 0] - tr[0, 4, 0]))
