#!/usr/bin/env python3
"""Measured deviation of the fused kernel from the float64 oracle on N(0,1) activations (the
distribution BASELINE's 'max-abs 1e-3' is meaningful for), per dtype, AdaIN on, train_input on."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from instantrestore_amd import ops
from oracle import shared_attn_oracle as O
torch.manual_seed(0)
B, H, L, N = 1, 2, 1024, 4
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ops.set_attn_variant(VARIANT)
C = H * 64
for dtype, name in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
    for scale_in, tag in ((1.0, "N(0,1)"), (0.25, "N(0,1/16)")):
        q, k, v = (torch.randn(B, L, C) * scale_in for _ in range(3))
        rk, rv = torch.randn(B, N, L, C) * scale_in, torch.randn(B, N, L, C) * scale_in
        q, k, v, rk, rv = (t.to(dtype) for t in (q, k, v, rk, rv))
        f = lambda t: t.float().numpy().astype(np.float64)
        ref = O.shared_attention_np(f(q), f(k), f(v), f(rk), f(rv), H, 0.125, True, True)
        c = lambda t: t.cuda()
        aff = ops.adain_stats(c(v), c(rv), heads=H)
        out = ops.shared_attention(c(q), c(k), c(v), c(rk), c(rv), heads=H, scale=0.125, include_self=True, adain=aff)
        err = np.abs(out.float().cpu().numpy() - ref)
        print(f"{name} {tag:10s}: max|O|={np.abs(ref).max():.4f}  max|err|={err.max():.2e}  mean|err|={err.mean():.2e}")
