"""device time of the AdaIN statistics calls at the three layer classes of cfg 2, replayed from a hipGraph of 40 calls (no launch
gaps): ir_adain_stats (self + N references), ir_token_stats over the references, ir_adain_stats_cached (self only).
usage: gpu_adain_time.py [B] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
REP = 40


def graph_time(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / REP * 1e3)
    return sorted(ts)[3]


for (L, C, H) in ((256, 1280, 20), (1024, 640, 10), (4096, 320, 5)):
    qkv_self = torch.randn(B, L, 3 * C, device="cuda").to(torch.bfloat16)
    qkv_ref = torch.randn(B * N, L, 3 * C, device="cuda").to(torch.bfloat16)
    v_self = qkv_self[..., 2 * C:]                         # the strided views the step hands over
    v_ref = qkv_ref[..., 2 * C:].reshape(B, N, L, C) if False else qkv_ref[..., 2 * C:].unflatten(0, (B, N))
    mb_ref, mb_self = B * N * L * C * 2 / 1e6, B * L * C * 2 / 1e6
    t = graph_time(lambda: ops.adain_stats(v_self, v_ref, heads=H))
    print(f"L={L:5d} C={C:5d}: adain_stats        {t:7.1f} us  {(mb_ref + mb_self) / t:6.2f} TB/s ({mb_ref + mb_self:.0f} MB)")
    t = graph_time(lambda: ops.token_stats(v_ref.flatten(0, 1).unsqueeze(1), heads=H))
    print(f"                  token_stats(refs)  {t:7.1f} us  {mb_ref / t:6.2f} TB/s ({mb_ref:.0f} MB)")
    m, sd = ops.token_stats(v_ref.flatten(0, 1).unsqueeze(1), heads=H)
    m, sd = m[:, 0].unflatten(0, (B, N)).contiguous(), sd[:, 0].unflatten(0, (B, N)).contiguous()
    t = graph_time(lambda: ops.adain_stats_cached(v_self, m, sd, heads=H))
    print(f"                  adain_stats_cached {t:7.1f} us  {mb_self / t:6.2f} TB/s ({mb_self:.0f} MB)")
