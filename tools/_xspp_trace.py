"""phase stamps of linear_xs_pp (debug build -DXSPP_TRACE=<block>): per chunk, group A (waves 0-3) and group B: start M | MFMAs done |
M confirmed (B) -> barrier | L start | L work done | confirmed (A) -> barrier.  usage: IR_LIB_PATH=... _xspp_trace.py [N] [fp32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
ops.LIN_KERNELS = {**ops.LIN_KERNELS, **ops.LIN_KERNELS_DEV}   # ids 9 / 10 exist in development builds (IR_LIB_PATH)
NS = int(os.environ.get("XSPP_STAMPS", "6"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 960
f32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
M, K = 131072, 320
x = torch.randn(M, K, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
for _ in range(3):
    y = ops.linear(x, w, kernel=ops.LIN_KERNELS["x_stationary_pp"])
torch.cuda.synchronize()
import numpy as np
blk = y.view(M // 512, 512 * N)[:, :16].contiguous().view(torch.int64).cpu().numpy()
blk = np.delete(blk, 100, axis=0)
e0 = blk[:, 0].min()
print("per block (100 MHz ticks = 10 ns): entry-min(entry)  X-load phase  loop   end-min(entry)   | by XCC")
for x in range(8):
    b = blk[(blk[:, 3] & 0xf) == x]
    if len(b):
        print(f"  xcc {x}: n={len(b):3d} entry {np.mean(b[:,0]-e0):6.0f}  xload {np.mean(b[:,1]-b[:,0]):6.0f} [{(b[:,1]-b[:,0]).min()}..{(b[:,1]-b[:,0]).max()}]  loop {np.mean(b[:,2]-b[:,1]):6.0f} [{(b[:,2]-b[:,1]).min()}..{(b[:,2]-b[:,1]).max()}]  end {np.mean(b[:,2]-e0):6.0f} max {(b[:,2]-e0).max()}")
raw = y.view(-1)[:2048].view(torch.int64).cpu().numpy()   # 512 stamps
A, B = raw[:256], raw[256:512]
t0 = min(A[0], B[0])
print("calibration: s_memtime ticks %d over s_memrealtime ticks %d (100 MHz) -> %.3f memtime ticks per ns" % (A[253] - A[252], A[255] - A[254], (A[253] - A[252]) / ((A[255] - A[254]) * 10.0)))
names = ["M0", "M1", "Mc", "L0", "L1", "Lc"]
print("stamps per chunk: start M, MFMAs done, (B: confirmed) at barrier, L start (after barrier), L work done, at barrier; units = s_memtime ticks")
for g, T in (("A", A), ("B", B)):
    print("group", g)
    for i in range(0, 255 // NS):
        r = T[NS * i:NS * i + NS + 1] - t0
        if NS == 6:   # start M, MFMAs done, at barrier, after barrier (L start), L work done, at barrier, next start
            print(f"  chunk {i:2d}: start {r[0]:7d}  M work {r[1]-r[0]:5d} confirm {r[2]-r[1]:5d} barrier {r[3]-r[2]:5d} | L work {r[4]-r[3]:5d} confirm {r[5]-r[4]:5d} barrier {r[6]-r[5]:5d}")
        else:         # ... L start, DMA issued, early stores, staged, late stores, at barrier
            print(f"  chunk {i:2d}: start {r[0]:7d}  M work {r[1]-r[0]:5d} confirm {r[2]-r[1]:5d} barrier {r[3]-r[2]:5d} | DMA {r[4]-r[3]:5d} stores(prev pair) {r[5]-r[4]:5d} stage {r[6]-r[5]:5d} stores {r[7]-r[6]:5d} confirm {r[8]-r[7]:5d} barrier {r[9]-r[8]:5d}")
