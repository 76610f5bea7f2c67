#!/usr/bin/env python3
"""Randomised check of ir_linear_fwd_stats over shapes (round 4): M whole 64-row blocks (including row counts that are not whole
workgroups / tiles), every K family, fp32 and 16-bit activations, with and without bias and a scaled leading third.  Per case:
Y bit-identical to the call without statistics, the partials merge to the float64 statistics of the rounded V within 1e-5,
and a canary behind the workspace survives.  usage: gpu_stats_fuzz.py [cases] [seed]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from instantrestore_amd import _lib, ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = _lib.lib()
bad = 0
for it in range(cases):
    K = int(rng.choice([64, 128, 192, 256, 320, 640, 1280]))
    H = int(rng.choice([1, 2, 5, 10, 20]))
    Cc = 64 * H
    N = 3 * Cc
    big = rng.random() < 0.25
    blocks = int(rng.integers(1024, 1100)) if big else int(rng.integers(1, 200))
    if big and K * blocks * 64 * 4 > 600e6:
        blocks = 1028
    M = 64 * blocks
    dt = torch.bfloat16 if rng.random() < 0.5 else torch.float16
    x32 = rng.random() < 0.5
    use_bias = rng.random() < 0.3
    scaled = rng.random() < 0.5
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    w[2 * Cc:] += (torch.randn(Cc, 1, generator=g) * 0.1).to(dt).cuda()
    x = torch.randn(M, K, generator=g)
    x = (x if x32 else x.to(dt)).cuda()
    b = (torch.randn(N, generator=g) * 0.2).to(dt).cuda() if use_bias else None
    if not ops.linear_supported(x, w, b):
        continue
    rows = ops.linear_stats_rows(M, N, K, use_bias)
    if rows != 64:
        print("skip (no statistics tail for this shape)", M, N, K, use_bias)
        continue
    need = (M // 64) * H * 128
    ws = torch.full((need + 4096,), 777.0, dtype=torch.float32, device="cuda")
    y = torch.empty(M, N, dtype=dt, device="cuda")
    kw = dict(scale_cols=Cc, col_scale=0.3) if scaled else {}
    rc = L.ir_linear_fwd_stats(0 if dt == torch.float16 else 1, 1 if x32 else 0, M, N, K, x.data_ptr(), K, w.data_ptr(), K,
                               None if b is None else b.data_ptr(), y.data_ptr(), N, Cc if scaled else 0, 0.3 if scaled else 1.0,
                               2 * Cc, Cc, ws.data_ptr(), need * 4, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    tag = f"M={M} N={N} K={K} {str(dt)[6:]} x32={x32} bias={use_bias} scaled={scaled} kernel={ops.linear_kernel_for(M, N, K, use_bias)}"
    if rc != 0:
        print("FAIL rc", rc, L.ir_last_error_string(), tag); bad += 1; continue
    torch.cuda.synchronize()
    ok = True
    if not bool((ws[need:] == 777.0).all()):
        print("FAIL canary", tag); ok = False
    y0 = ops.linear(x, w, b, **kw)
    if not torch.equal(y0, y):
        print("FAIL y differs from the call without statistics", tag); ok = False
    st = ops.ColumnStats(ws[:need].view(M // 64, H, 128), 64, H)
    m1, s1 = ops.token_stats_from_partials(st, 1, M) if M // 64 <= ops.STATS_MAX_CHUNKS else (None, None)
    if m1 is None:     # more partials than one merge takes: merge per 64-row set instead (means only) and check the partial means
        v = y[:, 2 * Cc:].float().reshape(M // 64, 64, H, 64)
        pm = ws[:need].view(M // 64, H, 128)[..., :64]
        pq = ws[:need].view(M // 64, H, 128)[..., 64:]
        e1 = float((pm - v.mean(1)).abs().max() / max(1e-30, float(v.mean(1).abs().max())))
        m2ref = ((v - v.mean(1, keepdim=True)) ** 2).sum(1)
        e2 = float((pq - m2ref).abs().max() / max(1e-30, float(m2ref.abs().max())))
    else:
        v = y[:, 2 * Cc:].double().reshape(1, M, H, 64)
        e1 = float((m1.double() - v.mean(1)).abs().max() / max(1e-30, float(v.mean(1).abs().max())))
        e2 = float((s1.double() - v.std(1, unbiased=True)).abs().max() / max(1e-30, float(v.std(1, unbiased=True).abs().max())))
    if not (e1 <= 2e-5 and e2 <= 2e-5):
        print("FAIL statistics", e1, e2, tag); ok = False
    bad += (not ok)
    del x, w, y, ws, y0
print(f"stats fuzz: {cases} cases, {bad} failing")
sys.exit(1 if bad else 0)
