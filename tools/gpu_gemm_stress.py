#!/usr/bin/env python3
"""Race screen of the projection GEMMs (round 4): every kernel x shape launched ITERS times on the same operands, each
result compared bit for bit with the first and (once) with a float64 product; a mismatch prints where it sits in its tile
(a rare early LDS read - guide: 'rare wrong tiles that come and go with shape, schedule edits or memory load').
usage: gpu_gemm_stress.py [iters=200] [load=0|1: a second stream keeps the memory system busy]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from instantrestore_amd import ops

ITERS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
LOAD = len(sys.argv) > 2 and sys.argv[2] == "1"
SHAPES = [(8192, 1920, 640), (8192, 3840, 1280), (8192, 1280, 1280), (32768, 960, 320), (2048, 3840, 1280), (32768, 1920, 640), (8192, 640, 640)]
KERNELS = [k for k in os.environ.get("KERNELS", "auto,256x256,128x128,128x256,256x128,64x128").split(",")]
side = torch.cuda.Stream()
junk = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
bad_total = 0
for dtype in (torch.float16, torch.bfloat16):
    for (M, N, K) in SHAPES:
        for x32 in (False, True):
            g = torch.Generator().manual_seed(M * 7 + N)
            x = torch.randn(M, K, generator=g)
            x = (x if x32 else x.to(dtype)).cuda()
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
            ref = (x.double() @ w.double().T)
            for kn in KERNELS:
                kid = ops.LIN_KERNELS[kn]
                try:
                    first = ops.linear(x, w, kernel=kid).clone()
                except Exception as e:
                    continue
                tol = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}[dtype]
                e0 = ((first.double() - ref).abs() > tol * ref.abs().clamp(min=1.0))
                nbad = 0
                where = []
                if bool(e0.any()):
                    idx = e0.nonzero()[:6].tolist()
                    where.append(("vs float64", idx))
                for it in range(ITERS):
                    if LOAD and it % 4 == 0:
                        with torch.cuda.stream(side):
                            junk.add_(1)
                    y = ops.linear(x, w, kernel=kid)
                    if not torch.equal(y, first):
                        nbad += 1
                        if len(where) < 4:
                            d = (y != first).nonzero()
                            where.append((it, d.shape[0], d[:4].tolist()))
                torch.cuda.synchronize()
                bad_total += nbad + (1 if e0.any() else 0)
                flag = "" if not (nbad or e0.any()) else "   <<<<<< MISMATCH"
                print("%s M=%6d N=%5d K=%5d %s %-8s: %d / %d runs differ from the first%s %s" % (
                    str(dtype)[6:], M, N, K, "fp32x" if x32 else "lowpx", kn, nbad, ITERS, flag, where if where else ""), flush=True)
print("TOTAL mismatching runs:", bad_total)
