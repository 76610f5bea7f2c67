import sys, os, torch
sys.path.insert(0, os.getcwd())
from instantrestore_amd import ops
B, N, L, H, t = 8, 4, 4096, 5, 1
dt = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(3)
C = H * 64
q, k, v = (torch.randn(B, L, C, device="cuda", generator=g).to(dt) for _ in range(3))
rk, rv = (torch.randn(B, N, L, C, device="cuda", generator=g).to(dt) for _ in range(2))
_, lse = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=True, return_lse=True)
def timed(kern, iters=5):
    for _ in range(2): ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=True, kernel=kern)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=True, kernel=kern)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
print(os.environ.get("IR_LIB_PATH", "product"), " ".join(f"{kn} {timed(kn):.3f} ms" for kn in ("lines64", "lines64k128", "lines32")), flush=True)
