"""phase stamps of linear_xs_rot (debug build -DXSROT_TRACE=<block>): per chunk: start | window A done | window B done | confirmed -> barrier
usage: IR_LIB_PATH=... [ZERO=1] _xsrot_trace.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
ops.LIN_KERNELS = {**ops.LIN_KERNELS, **ops.LIN_KERNELS_DEV}   # ids 9 / 10 exist in development builds (IR_LIB_PATH)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
M, K = 131072, 320
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
if os.environ.get("ZERO"):
    x.zero_(); w.zero_()
for _ in range(int(os.environ.get("LAUNCHES", "3"))):
    y = ops.linear(x, w, kernel=ops.LIN_KERNELS["x_stationary_rot"])
torch.cuda.synchronize()
raw = y.view(-1)[:2048].view(torch.int64).cpu().numpy()
A = raw[:256]
print("launches %s: loop %d shader cycles in %.1f us -> %.3f GHz" % (os.environ.get("LAUNCHES", "3"), A[253] - A[252], (A[255] - A[254]) / 100.0, (A[253] - A[252]) / ((A[255] - A[254]) * 10.0)))
if os.environ.get("BRIEF"):
    sys.exit(0)
for g, T in (("waves 0-3", raw[:256]), ("waves 4-7", raw[256:512])):
    print(g)
    t0 = T[0]
    for i in range(0, 30):
        r = T[4 * i:4 * i + 5] - t0
        print(f"  chunk {i:2d}: start {r[0]:7d}  DMA+window A {r[1]-r[0]:5d}  window B {r[2]-r[1]:5d}  confirm {r[3]-r[2]:5d}  barrier {r[4]-r[3]:5d}")
