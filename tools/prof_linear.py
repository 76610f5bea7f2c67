"""rocprofv3 target: ir_linear_fwd on one shape.  usage: prof_linear.py [M N K [kernel [fp32]]] (kernel: ops.LIN_KERNELS name)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
ops.LIN_KERNELS = {**ops.LIN_KERNELS, **ops.LIN_KERNELS_DEV}   # ids 9 / 10 exist in development builds (IR_LIB_PATH)
M, N, K = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (131072, 960, 320)
kid = ops.LIN_KERNELS[sys.argv[4]] if len(sys.argv) > 4 else 0
f32 = len(sys.argv) > 5 and sys.argv[5] == "fp32"
x = torch.randn(M, K, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
for _ in range(5):
    ops.linear(x, w, kernel=kid)
torch.cuda.synchronize()
