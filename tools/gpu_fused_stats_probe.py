#!/usr/bin/env python3
"""What the statistics tail of the q/k/v GEMMs costs, shape by shape (round 4): ops.linear with and without stats=(2C, C),
interleaved sustained runs (SECS each), next to the standalone passes it replaces and the merges that remain.
usage: gpu_fused_stats_probe.py [cfg2|cfg4|cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from instantrestore_amd import ops
from instantrestore_amd.roofline import layer_classes

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B, N, px, dt, _ = bench.CONFIGS[cfg]
dtype = bench.DT[dt]
SECS = float(os.environ.get("SECS", "0.3"))


def sustained(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for (L, C, H) in layer_classes(px):
    w = (torch.randn(3 * C, C, device="cuda") / C ** 0.5).to(dtype)
    for sets, tag in ((B * N, "capture"), (B, "shared ")):
        x = torch.randn(sets, L, C, device="cuda")
        res = {}
        for rnd in range(3):
            res.setdefault("plain", []).append(sustained(lambda: ops.linear(x, w, None, scale_cols=C, col_scale=0.18)))
            res.setdefault("stats", []).append(sustained(lambda: ops.linear(x, w, None, scale_cols=C, col_scale=0.18, stats=(2 * C, C))))
        y, st = ops.linear(x, w, None, stats=(2 * C, C))
        v = y[..., 2 * C:]
        t_tok = sustained(lambda: ops.token_stats(v.unsqueeze(1), heads=H))
        t_tsp = sustained(lambda: ops.token_stats_from_partials(st, sets, L))
        line = "L=%5d C=%4d %s M=%6d | GEMM %7.1f us, with tail %7.1f (+%5.1f) | standalone token_stats %6.1f | merge of partials %5.1f" % (
            L, C, tag, sets * L, min(res["plain"]), min(res["stats"]), min(res["stats"]) - min(res["plain"]), t_tok, t_tsp)
        if tag.startswith("shared"):
            rv = torch.randn(B, N, L, C, device="cuda").to(dtype)
            cm, cs = ops.token_stats(rv.reshape(B * N, 1, L, C), heads=H)
            cm, cs = cm.reshape(B, N, H, 64).contiguous(), cs.reshape(B, N, H, 64).contiguous()
            t_full = sustained(lambda: ops.adain_stats(v, rv, heads=H))
            t_cached = sustained(lambda: ops.adain_stats_cached(v, cm, cs, heads=H))
            t_aff = sustained(lambda: ops.adain_affine_from_partials(st, B, L, N, L, content_mean=cm, content_std=cs))
            line += " | adain_stats %6.1f, cached %5.1f, affine from partials %5.1f" % (t_full, t_cached, t_aff)
        print(line, flush=True)
