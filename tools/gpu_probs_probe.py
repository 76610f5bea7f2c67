"""The probability-dump kernels (ir_attn_probs_ex) at the layer classes of a config: ms, GB/s of H*L*Lkv*2 bytes written,
fraction of the copy / fill rate measured in the same run on the same box, every kernel beside the others.
usage: python tools/gpu_probs_probe.py [B] [N] [px] [dtype]      (defaults: cfg 2 = 8 4 512 bf16)"""
import sys
import torch
sys.path.insert(0, ".")
from instantrestore_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
px = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
f = (px // 512) ** 2


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


print(f"device: {torch.cuda.get_device_name(0)}  B={B} N={N} px={px} {dtype}")
n = 1 << 30   # 2 GiB of 16-bit elements each
src = torch.randn(n // 4, device="cuda").to(dtype).repeat(4)
dst = torch.empty_like(src)
t_copy = timeit(lambda: dst.copy_(src))
t_fill = timeit(lambda: dst.fill_(0.5))
copy_rate = 2 * n * 2 / t_copy / 1e6   # read + write bytes
fill_rate = n * 2 / t_fill / 1e6
print(f"copy 2 GiB -> 2 GiB : {t_copy:7.3f} ms  {copy_rate:7.1f} GB/s (read + write)")
print(f"fill 2 GiB          : {t_fill:7.3f} ms  {fill_rate:7.1f} GB/s (write only)")
del src, dst

for (L, H) in ((256 * f, 20), (1024 * f, 10), (4096 * f, 5)):
    C = H * 64
    for t in (1, 0):
        g = torch.Generator(device="cuda").manual_seed(3)
        q = torch.randn(B, L, C, device="cuda", generator=g).to(dtype)
        k = torch.randn(B, L, C, device="cuda", generator=g).to(dtype)
        v = torch.randn(B, L, C, device="cuda", generator=g).to(dtype)
        rk = torch.randn(B, N, L, C, device="cuda", generator=g).to(dtype)
        rv = torch.randn(B, N, L, C, device="cuda", generator=g).to(dtype)
        inc = bool(t)
        _, lse = ops.shared_attention(q, k, v, rk, rv, heads=H, scale=0.125, include_self=inc, return_lse=True)
        lkv = (N + t) * L
        nbytes = B * H * L * lkv * 2
        if nbytes > 40e9:
            print(f"L={L} t={t}: {nbytes / 1e9:.1f} GB of probabilities - skipped")
            continue
        ref = None
        line = f"L={L:5d} H={H:2d} t={t} Lkv={lkv:6d} {nbytes / 1e9:6.2f} GB :"
        for kern in ("generic", "lines64", "lines32", "lines32k128", "lines64k128", "lines32k256"):
            iters = 3 if nbytes > 2e9 else 10
            ms = timeit(lambda: ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=inc, kernel=kern), iters=iters, warm=1)
            p = ops.attn_probs(q, k, rk, lse, heads=H, scale=0.125, include_self=inc, kernel=kern)
            if ref is None:
                ref = p
                same = ""
            else:
                same = " =" if torch.equal(p, ref) else " DIFFERS"
            del p
            rate = nbytes / ms / 1e6
            line += f"  {kern} {ms:8.3f} ms {rate:7.1f} GB/s ({rate / fill_rate:4.2f} of fill, {rate / (copy_rate):4.2f} of copy){same}"
        ms = timeit(lambda: ops.attn_segment_mass(q, k, rk, lse, heads=H, scale=0.125, include_self=inc), iters=5, warm=1)
        line += f"  | segment mass {ms:7.3f} ms"
        print(line, flush=True)
        del ref
