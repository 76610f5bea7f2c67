#!/usr/bin/env python3
"""On-GPU diagnostics for the fused attention kernel: one-hot attention reveals any key / channel
permutation error in the MFMA operand layouts (development aid; not part of the test suite)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops


def onehot_case(dtype, L, N, Lr, inc, variant):
    torch.manual_seed(0)
    B, H = 1, 1
    lkv = (L if inc else 0) + N * Lr
    kall = torch.randn(lkv, 64)
    pi = torch.randperm(lkv)[:L] if lkv >= L else torch.randint(0, lkv, (L,))
    q = (6.0 * kall[pi]).reshape(1, L, 64)
    vall = (torch.arange(lkv).float()[:, None] % 512) + torch.arange(64).float()[None, :] / 128.0
    k_self = kall[:L].reshape(1, L, 64) if inc else torch.zeros(1, L, 64)
    v_self = vall[:L].reshape(1, L, 64) if inc else torch.zeros(1, L, 64)
    off = L if inc else 0
    rk = kall[off:].reshape(1, N, Lr, 64) if N else None
    rv = vall[off:].reshape(1, N, Lr, 64) if N else None
    c = lambda t: None if t is None else t.to(dtype).cuda()
    ops.set_attn_variant(variant)
    presc = variant == 16        # the 128-row kernel (round 6) takes pre-scaled Q only
    if presc:
        q = q * (0.125 * 1.4426950408889634)
    out = ops.shared_attention(c(q), c(k_self), c(v_self), c(rk), c(rv), heads=1, scale=0.125, include_self=inc, q_prescaled=presc)
    torch.cuda.synchronize()
    out = out.float().cpu()[0]
    want = vall[pi].to(dtype).float()
    err = (out - want).abs()
    bad = (err.max(dim=1).values > 0.51).nonzero().flatten()
    print(f"  onehot dtype={dtype} L={L} N={N} Lr={Lr} inc={inc} v={variant}: max err {err.max():.3f}, bad rows {len(bad)}/{L}")
    for i in bad[:6].tolist():
        print(f"    row {i}: want key {pi[i].item()} got ~{out[i,0].item():.2f}  d-pattern {[round(x,3) for x in (out[i,:6]-out[i,0]).tolist()]}")
    return len(bad) == 0


def main():
    print(torch.cuda.get_device_name(0), ops._lib.lib().ir_build_info().decode())
    ok = True
    for variant in (0, 1, 2, 3, 4, 6, 7, 8, 9, 12):
        for dtype in (torch.float16, torch.bfloat16):
            ok &= onehot_case(dtype, 64, 0, 0, True, variant)
            ok &= onehot_case(dtype, 256, 2, 128, True, variant)
            ok &= onehot_case(dtype, 100, 3, 72, False, variant)
    print("DIAG", "OK" if ok else "FAILED")


if __name__ == "__main__":
    main()
