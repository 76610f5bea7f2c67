"""debug aid for the 128-row kernel: small shapes, error maps against the 64-row kernel and the fp64 oracle.
usage: gpu_w128_debug.py [L ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from instantrestore_amd import ops
from oracle import shared_attn_oracle as O
C_ = 0.125 * 1.4426950408889634
dt = torch.bfloat16
for L in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    torch.manual_seed(L)
    q = (torch.randn(1, L, 64, device="cuda") * C_).to(dt)
    k, v = torch.randn(1, L, 64, device="cuda").to(dt), (torch.randn(1, L, 64, device="cuda") * 0.9 + 0.3).to(dt)
    if os.environ.get("VMODE") == "ch":      # V[key][d] = d: any P gives O[d] = d
        v = torch.arange(64, device="cuda").float().view(1, 1, 64).expand(1, L, 64).contiguous().to(dt)
    if os.environ.get("VMODE") == "key":     # V[key][d] = key / 64: O = E[key]
        v = (torch.arange(L, device="cuda").float() / 64).view(1, L, 1).expand(1, L, 64).contiguous().to(dt)
    res = {}
    for var in (13, 16):
        ops.set_attn_variant(var)
        out, lse = ops.shared_attention(q, k, v, heads=1, scale=0.125, include_self=True, q_prescaled=True, return_lse=True, out_dtype=torch.float32)
        torch.cuda.synchronize()
        res[var] = (out.float().cpu().numpy()[0], lse.cpu().numpy()[0, 0])
    ops.set_attn_variant(0)
    f = lambda t: t.float().cpu().numpy().astype(np.float64)
    ref = O.shared_attention_np(f(q) / C_, f(k), f(v), None, None, 1, 0.125, False, True)[0]
    o13, l13 = res[13]; o16, l16 = res[16]
    print(f"L={L}: |w64-ref| {np.abs(o13-ref).max():.3e}  |w128-ref| {np.abs(o16-ref).max():.3e}  lse diff {np.abs(l13-l16).max():.3e}  finite {np.isfinite(o16).all()}")
    e = np.abs(o16 - ref)
    nb = (L + 31) // 32
    print("  err by 32-row block x 8-channel group (max):")
    for rb in range(nb):
        print("   rows %4d+: " % (32 * rb) + " ".join("%8.1e" % e[32 * rb:32 * rb + 32, 8 * c:8 * c + 8].max() for c in range(8)),
              " lse err %.2e" % np.abs(l13 - l16)[32 * rb:32 * rb + 32].max())
    r = 0
    print("  row 0 ref :", np.round(ref[r, :8], 3), "\n  row 0 w128:", np.round(o16[r, :8], 3))
