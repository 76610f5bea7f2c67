"""rocprofv3 target: ir_linear_fwd at the fused-QKV shape of the 64x64-token layer class."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from instantrestore_amd import ops
M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (131072, 960)
x = torch.randn(M, 320, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, 320, device="cuda", dtype=torch.bfloat16)
for _ in range(5):
    ops.linear(x, w)
torch.cuda.synchronize()
