"""W64 (variant 12) vs the 32-row pipelined kernel (variant 10) on shapes with and without tail
quantisation of the item grid (no AdaIN, no split): what 64 rows per wave buys per item."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from instantrestore_amd import ops
from instantrestore_amd.roofline import attn_flops
dt = torch.bfloat16
N, L = 4, 4096
for B, H in ((8, 4), (8, 5), (16, 4), (4, 4), (32, 5)):
    C = H * 64
    Nn = N if B < 32 else 0
    q, k, v = (torch.randn(B, L, C, device="cuda").to(dt) for _ in range(3))
    rk = torch.randn(B, Nn, L, C, device="cuda").to(dt) if Nn else None
    rv = torch.randn(B, Nn, L, C, device="cuda").to(dt) if Nn else None
    for var in (10, 12, 10, 12):
        ops.set_attn_variant(var)
        for split in (True, False):
            ops.time_shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, iters=3, split=split)
            ms = min(ops.time_shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, iters=10, split=split) for _ in range(3))
            print(f"B{B} H{H} N{Nn} v{var} split={int(split)}: {ms:.4f} ms {attn_flops(B, L, (Nn+1)*L, C)/ms/1e9:7.1f} TF/s  items128={B*H*32} items256={B*H*16}")
