#!/usr/bin/env python3
"""Development aid: one kernel variant against the float64 oracle on a few shapes, error localised by (batch, head, 32-row
block, 32-channel half).  usage: gpu_sp_debug.py [variant=16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from instantrestore_amd import ops
from oracle import shared_attn_oracle as O

var = int(sys.argv[1]) if len(sys.argv) > 1 else 16
f = lambda t: None if t is None else t.float().cpu().numpy().astype(np.float64)
for (B, H, L, N, Lr, inc, ad, dt) in [(1, 1, 64, 0, 0, True, False, torch.bfloat16), (1, 1, 64, 1, 64, True, False, torch.bfloat16),
                                      (2, 2, 64, 4, 64, True, False, torch.float16), (1, 2, 256, 4, 256, True, True, torch.bfloat16),
                                      (1, 1, 300, 3, 200, False, True, torch.bfloat16), (1, 5, 1024, 4, 1024, True, True, torch.bfloat16)]:
    g = torch.Generator().manual_seed(5)
    C = H * 64
    q, k, v = (torch.randn(B, L, C, generator=g).to(dt) for _ in range(3))
    rk = torch.randn(B, N, Lr, C, generator=g).to(dt) if N else None
    rv = (torch.randn(B, N, Lr, C, generator=g) * 1.3 - 0.4).to(dt) if N else None
    ref = O.shared_attention_np(f(q), f(k), f(v), f(rk), f(rv), H, 0.125, ad, inc)
    c = lambda t: None if t is None else t.cuda()
    ops.set_attn_variant(var)
    aff = ops.adain_stats(c(v), c(rv), heads=H) if ad else None
    out = ops.shared_attention(c(q), c(k), c(v), c(rk), c(rv), heads=H, scale=0.125, include_self=inc, adain=aff)
    torch.cuda.synchronize()
    o = out.float().cpu().numpy().astype(np.float64)
    err = np.abs(o - ref)
    bad = ~np.isfinite(o)
    print(f"B{B} H{H} L{L} N{N} Lr{Lr} inc{inc} ad{ad} {dt}: max err {np.nanmax(err):.3e} nonfinite {bad.sum()} max|ref| {np.abs(ref).max():.3f}")
    if bad.any() or np.nanmax(err) > 8e-3 * max(1, np.abs(ref).max()):
        e = np.where(bad, 1e9, err).reshape(B, -1, H, 64)
        for b in range(B):
            for h in range(H):
                for r0 in range(0, L, 32):
                    blk = e[b, r0:r0 + 32, h]
                    if blk.max() > 8e-3:
                        rows = np.where(blk.max(1) > 8e-3)[0]
                        cols = np.where(blk.max(0) > 8e-3)[0]
                        print(f"   b{b} h{h} rows {r0}+{rows.min()}..{rows.max()} ({len(rows)}) cols {cols.min()}..{cols.max()} ({len(cols)}) max {blk.max():.3e}")
ops.set_attn_variant(0)
