#!/usr/bin/env python3
"""Board power and shader clock while the fused attention kernel runs back to back (development aid).
usage: gpu_power_probe.py [variants=13,15] [seconds=3]   - polls rocm-smi from a thread during the run."""
import sys, os, subprocess, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops

variants = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "13").split(",")]
PRESC = len(sys.argv) > 3 and sys.argv[3] == "presc"   # also run every variant with IR_FLAG_Q_PRESCALED
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
B, N, L, H = 8, 4, 4096, 5
C = H * 64
torch.manual_seed(0)
dt = torch.bfloat16
q, k, v = (torch.randn(B, L, C, device="cuda").to(dt) for _ in range(3))
rk = torch.randn(B, N, L, C, device="cuda").to(dt)
rv = torch.randn(B, N, L, C, device="cuda").to(dt)
aff = ops.adain_stats(v, rv, heads=H)

def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", r)
            sc = re.findall(r"sclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", r)
            out.append((pw[0] if pw else "?", sc[0] if sc else "?"))
        except Exception as e:  # noqa
            out.append(("err", str(e)[:40]))
        time.sleep(0.3)

def run(var, data, presc=False):
    ops.set_attn_variant(var)
    qq, kk, vv, rkk, rvv = data
    if presc:
        qq = (qq.float() * (0.125 * 1.4426950408889634)).to(qq.dtype)
    kw = dict(heads=H, scale=0.125, include_self=True, adain=aff, q_prescaled=presc)
    ops.shared_attention(qq, kk, vv, rkk, rvv, **kw)
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, out)); th.start()
    t0 = time.time(); n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            ops.shared_attention(qq, kk, vv, rkk, rvv, **kw)
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    ms = e0.elapsed_time(e1) / n
    return ms, out

zeros = tuple(torch.zeros_like(x) for x in (q, k, v, rk, rv))
for var in variants:
    for name, data in (("random", (q, k, v, rk, rv)), ("zeros", zeros)):
        for presc in ((False, True) if PRESC else (False,)):
            ms, out = run(var, data, presc)
            print(f"v{var} {name}{' prescaled-Q' if presc else ''}: {ms:.4f} ms/launch  samples (W, sclk MHz): {out[1:-1][:8]}")
print(subprocess.run(["rocm-smi", "--showmaxpower", "-d", "0"], capture_output=True, text=True).stdout[-300:])
