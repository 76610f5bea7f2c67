import sys, os
sys.path.insert(0, "/root/repo")
import torch
from instantrestore_amd import ops
from instantrestore_amd.roofline import attn_flops
B, N, L, H = 8, 4, 4096, 5
C = H * 64
dt = torch.bfloat16
for name, fill in (("random", None), ("zeros", 0.0), ("const", 0.37), ("random again", None)):
    if fill is None:
        q, k, v = (torch.randn(B, L, C, device="cuda").to(dt) for _ in range(3))
        rk = torch.randn(B, N, L, C, device="cuda").to(dt); rv = torch.randn(B, N, L, C, device="cuda").to(dt)
    else:
        q, k, v = (torch.full((B, L, C), fill, device="cuda", dtype=dt) for _ in range(3))
        rk = torch.full((B, N, L, C), fill, device="cuda", dtype=dt); rv = rk.clone()
    for var in (10, 0, 10, 0):
        ops.set_attn_variant(var)
        aff = ops.adain_stats(v, rv, heads=H)
        ops.time_shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, iters=3, adain=aff)
        ms = min(ops.time_shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, iters=10, adain=aff) for _ in range(3))
        print(f"{name:13s} v{var}: {ms:.4f} ms {attn_flops(B, L, 5*L, C)/ms/1e9:7.1f} TF/s")
