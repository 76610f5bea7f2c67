"""rocprofv3 target: ir_linear_fwd on one shape.  usage: prof_linear.py [M N K [kernel [fp32 [stats]]]] (kernel: ops.LIN_KERNELS name;
stats: N = 3C, the token statistics of the last third ride in the GEMM - kernel must be auto)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
M, N, K = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (131072, 960, 320)
kid = ops.LIN_KERNELS[sys.argv[4]] if len(sys.argv) > 4 else 0
f32 = len(sys.argv) > 5 and sys.argv[5] == "fp32"
x = torch.randn(M, K, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
st = len(sys.argv) > 6 and sys.argv[6] == "stats"
for _ in range(5):
    if st:
        ops.linear(x, w, stats=(2 * N // 3, N // 3))
    else:
        ops.linear(x, w, kernel=kid)
torch.cuda.synchronize()
