#!/usr/bin/env python3
"""Kernel-only timing of the fused shared attention at the real layer classes (development aid).
usage: gpu_time.py [variants=0,1,2] [B=8] [N=4] [px=512] [presc]   (presc: q pre-scaled, IR_FLAG_Q_PRESCALED - what the processors launch)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
from instantrestore_amd.roofline import layer_classes, attn_flops

variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4
px = int(sys.argv[4]) if len(sys.argv) > 4 else 512
PRESC = len(sys.argv) > 5 and sys.argv[5] == "presc"
QC = 0.125 * 1.4426950408889634 if PRESC else 1.0
dtype = torch.bfloat16
torch.manual_seed(0)
for (L, C, H) in layer_classes(px):
    q, k, v = (torch.randn(B, L, C, device="cuda").to(dtype) for _ in range(3))
    q = (q.float() * QC).to(dtype)
    rk = torch.randn(B, N, L, C, device="cuda").to(dtype)
    rv = torch.randn(B, N, L, C, device="cuda").to(dtype)
    aff = ops.adain_stats(v, rv, heads=H)
    for t in (1, 0):
        for ad in (True, False):
            row = []
            for var in variants:
                ops.set_attn_variant(var)
                kw = dict(heads=H, scale=0.125, include_self=bool(t), adain=aff if ad else None, q_prescaled=PRESC)
                ops.time_shared_attention(q, k, v, rk, rv, iters=2, **kw)
                ms = min(ops.time_shared_attention(q, k, v, rk, rv, iters=10, **kw) for _ in range(3))
                tf = attn_flops(B, L, (N + t) * L, C) / ms / 1e9
                row.append(f"v{var}: {ms:8.4f} ms {tf:7.1f} TF/s")
            print(f"L={L:5d} H={H:2d} t={t} adain={int(ad)} | " + " | ".join(row), flush=True)
    # plain self attention over the B*N reference token sets (K/V capture shape)
    qq = torch.randn(B * N, L, C, device="cuda").to(dtype)
    qs = (qq.float() * QC).to(dtype)
    row = []
    for var in variants:
        ops.set_attn_variant(var)
        ops.time_shared_attention(qs, qq, qq, iters=2, heads=H, scale=0.125, q_prescaled=PRESC)
        ms = min(ops.time_shared_attention(qs, qq, qq, iters=10, heads=H, scale=0.125, q_prescaled=PRESC) for _ in range(3))
        row.append(f"v{var}: {ms:8.4f} ms {attn_flops(B * N, L, L, C) / ms / 1e9:7.1f} TF/s")
    print(f"L={L:5d} H={H:2d} plain self-attn x{B*N} | " + " | ".join(row), flush=True)
ops.set_attn_variant(0)

# AdaIN statistics (HBM-streaming kernel): achieved GB/s against the one-pass byte count
print("--- adain_stats (one pass over V_self and the N reference V) ---")
for (L, C, H) in layer_classes(px):
    v = torch.randn(B, L, C, device="cuda").to(dtype)
    rv = torch.randn(B, N, L, C, device="cuda").to(dtype)
    for _ in range(3):
        ops.adain_stats(v, rv, heads=H)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.adain_stats(v, rv, heads=H)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = 2.0 * B * C * (L + N * L)
    print(f"L={L:5d} C={C:4d}: {ms*1e3:7.1f} us  {nbytes/1e6:7.1f} MB  {nbytes/ms/1e6:7.1f} GB/s  ({nbytes/ms/1e6/8000*100:4.1f}% of 8 TB/s)")
