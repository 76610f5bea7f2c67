#!/usr/bin/env python3
"""What the per-segment attention mass costs (ABI v9 ``seg_mass``): per cfg-2 layer class, bf16, pre-scaled Q, AdaIN fold, t = 1 -
the attention launch alone, the same launch with the masses as a by-product (MASS instantiation + the row-sized finishing kernel),
and the round's first form: launch with LSE + the second pass ``ir_attn_segment_mass`` over Q and K.
usage: python tools/gpu_seg_mass_time.py [B=8] [N=4] [iters=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 30
QC = 0.125 * 1.4426950408889634
dt = torch.bfloat16


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"# per-segment attention mass: cost per layer class (B={B}, N={N}, t=1, AdaIN fold, bf16, pre-scaled Q; ms per call, {iters} calls back to back)")
print("# L | H | attention alone | + masses as a by-product | delta | attention with LSE + second pass (ir_attn_segment_mass) | delta | max |by-product - second pass|")
for L, H in ((256, 20), (1024, 10), (4096, 5)):
    C = H * 64
    torch.manual_seed(L)
    q = (torch.randn(B, L, C, device="cuda") * QC).to(dt)
    k, v = torch.randn(B, L, C, device="cuda").to(dt), torch.randn(B, L, C, device="cuda").to(dt)
    rk, rv = torch.randn(B, N, L, C, device="cuda").to(dt), torch.randn(B, N, L, C, device="cuda").to(dt)
    aff = ops.adain_stats(v, rv, heads=H)
    kw = dict(heads=H, scale=0.125, include_self=True, adain=aff, q_prescaled=True)
    t0 = timed(lambda: ops.shared_attention(q, k, v, rk, rv, **kw), iters)
    t1 = timed(lambda: ops.shared_attention(q, k, v, rk, rv, return_mass=True, **kw), iters)

    def second():
        _, lse = ops.shared_attention(q, k, v, rk, rv, return_lse=True, **kw)
        return ops.attn_segment_mass(q, k, rk, lse, heads=H, scale=0.125, include_self=True, q_prescaled=True)
    t2 = timed(second, iters)
    m1 = ops.shared_attention(q, k, v, rk, rv, return_mass=True, **kw)[1]
    d = float((m1 - second()).abs().max())
    print(f"{L:5d} | {H:2d} | {t0:7.4f} | {t1:7.4f} | {t1 - t0:+7.4f} | {t2:7.4f} | {t2 - t0:+7.4f} | {d:.2e}", flush=True)
