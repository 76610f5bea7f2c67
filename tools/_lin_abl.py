"""timing of one tiled-GEMM config under the ablation builds (IR_LIB_PATH); usage: _lin_abl.py M N K kernel [fp32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
ops.LIN_KERNELS = {**ops.LIN_KERNELS, **ops.LIN_KERNELS_DEV}   # ids 9 / 10 exist in development builds (IR_LIB_PATH)
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kid = ops.LIN_KERNELS[sys.argv[4]]
f32 = len(sys.argv) > 5 and sys.argv[5] == "fp32"
x = torch.randn(M, K, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
if os.environ.get("ZERO"):   # all-zero operands: same instruction stream and traffic, far less switching power
    x.zero_(); w.zero_()
for _ in range(5): ops.linear(x, w, kernel=kid)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, w, kernel=kid)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print("ZERO" if os.environ.get("ZERO") else "rand", os.environ.get("IR_LIB_PATH", "product"), sys.argv[1:], "%.1f us" % sorted(ts)[2], "%.0f TF/s" % (2.0 * M * N * K / sorted(ts)[2] / 1e6))
