"""one kernel of the probability dump at the cfg-2 top layer, a few launches: the target of rocprofv3 passes
usage: python tools/prof_probs.py [kernel=lines64] [L=4096] [H=5] [B=8] [N=4] [t=1]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instantrestore_amd import ops
kern = sys.argv[1] if len(sys.argv) > 1 else "lines64"
L, H, B, N, t = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 4096), (3, 5), (4, 8), (5, 4), (6, 1)))
dtype = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(3)
C = H * 64
q, k, v = (torch.randn(B, L, C, device="cuda", generator=g).to(dtype) for _ in range(3))
rk, rv = (torch.randn(B, N, L, C, device="cuda", generator=g).to(dtype) for _ in range(2))
_, lse = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=bool(t), return_lse=True)
for _ in range(4):
    if kern == "mass":
        m = ops.attn_segment_mass(q, k, rk, lse, heads=H, scale=0.125, include_self=bool(t))
    else:
        p = ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=bool(t), kernel=kern)
torch.cuda.synchronize()
