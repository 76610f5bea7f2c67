#!/usr/bin/env python3
"""Within-process A/B of kernel ablations (interleaved rounds, min and median reported).
Needs an ablation build of the library: `instantrestore_amd/csrc/build.sh -DIR_ABLATIONS`.
ABL bits: 1 = no staging/barrier, 2 = no softmax max/exp, 4 = no LDS fragment reads."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
from instantrestore_amd.roofline import attn_flops
B, N, L, H = 8, 4, 4096, 5
C = H * 64
torch.manual_seed(0)
dt = torch.bfloat16
q, k, v = (torch.randn(B, L, C, device="cuda").to(dt) for _ in range(3))
rk = torch.randn(B, N, L, C, device="cuda").to(dt); rv = torch.randn(B, N, L, C, device="cuda").to(dt)
variants = [2] + [2 | (a << 5) for a in range(1, 8)]   # tuning value: kernel in bits 0-4, ablation bits above
if len(sys.argv) > 1: variants = [int(x, 0) for x in sys.argv[1].split(",")]
res = {v_: [] for v_ in variants}
kw = dict(heads=H, scale=0.125, include_self=True)
for rnd in range(5):
    for var in variants:
        ops.set_attn_variant(var)
        res[var].append(ops.time_shared_attention(q, k, v, rk, rv, iters=5, **kw))
fl = attn_flops(B, L, 5 * L, C)
for var in variants:
    ms = res[var]
    print(f"variant 0x{var:02x} abl={var>>5}: min {min(ms):.4f} ms  med {statistics.median(ms):.4f} ms  -> {fl/min(ms)/1e9:7.1f} TF/s(min)")
ops.set_attn_variant(0)
