#!/usr/bin/env python3
"""Run only the fused shared-attention kernel (top layer class of cfg2) a few times: the target
of `rocprofv3 --pmc ...` counter passes.  usage: prof_attn.py [variant] [iters] [L] [H] [t] [adain] [presc]
presc = 1: q is handed over pre-scaled (IR_FLAG_Q_PRESCALED), the way the processors produce it at this shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
var = int(sys.argv[1]) if len(sys.argv) > 1 else 0
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
L = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
H = int(sys.argv[4]) if len(sys.argv) > 4 else 5
t = int(sys.argv[5]) if len(sys.argv) > 5 else 1
ad = int(sys.argv[6]) if len(sys.argv) > 6 else 1
presc = int(sys.argv[7]) if len(sys.argv) > 7 else 0
B, N, C = 8, 4, H * 64
torch.manual_seed(0)
dtype = torch.bfloat16
q, k, v = (torch.randn(B, L, C, device="cuda").to(dtype) for _ in range(3))
rk = torch.randn(B, N, L, C, device="cuda").to(dtype)
rv = torch.randn(B, N, L, C, device="cuda").to(dtype)
aff = ops.adain_stats(v, rv, heads=H) if ad else None
ops.set_attn_variant(var)
if presc:
    q = (q.float() * (0.125 * 1.4426950408889634)).to(dtype)
for _ in range(iters):
    out = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=bool(t), adain=aff, q_prescaled=bool(presc))
torch.cuda.synchronize()
print("done", float(out.float().abs().mean()))
