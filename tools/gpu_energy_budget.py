#!/usr/bin/env python3
"""Energy budget of the dominant kernel: joules per launch of the 64-row QS kernel and of its ABLATIONS
(development build: tools/experiments/build.sh; IR_LIB_PATH=tools/experiments/libinstantrestore_hip_dev.so), same method as
tools/gpu_energy_probe.py - every row runs back to back for SECS seconds on N(0,1) operands while rocm-smi is sampled from a
side thread (first sample dropped), J/launch = average board W x sustained ms.

Two readings of the same launch:
  ladder         each row ADDS one class of work to the row before it: matrix skeleton (MFMAs + K / V fragment reads from LDS)
                 -> + exponentials -> + row sums -> + fp32->16-bit conversions -> + Q-fragment re-reads from LDS -> + LDS-DMA and
                 the per-tile barrier = the product kernel.  The increments telescope to the product kernel's joules.
  leave-one-out  the product kernel with ONE class removed: what removing that class alone would buy.
The ablated kernels compute WRONG results (that is what an ablation is); their operands are the same random tensors.
usage: IR_LIB_PATH=... python tools/gpu_energy_budget.py [shape=top|capture] [secs=2.5] [idle_secs=3]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
from instantrestore_amd.roofline import attn_flops

shape = sys.argv[1] if len(sys.argv) > 1 else "top"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
idle_secs = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
B, N, L, H = 8, 4, 4096, 5
shared = shape == "top"
C = H * 64
dt = torch.bfloat16
QC = 0.125 * 1.4426950408889634
torch.manual_seed(0)
S = B if shared else B * N
q, k, v = (torch.randn(S, L, C, device="cuda").to(dt) for _ in range(3))
q = (q.float() * QC).to(dt)
if shared:
    rk, rv = torch.randn(B, N, L, C, device="cuda").to(dt), torch.randn(B, N, L, C, device="cuda").to(dt)
    aff = ops.adain_stats(v, rv, heads=H)
    args, kw, flops = (q, k, v, rk, rv), dict(heads=H, scale=0.125, include_self=True, adain=aff, q_prescaled=True), attn_flops(B, L, 5 * L, C)
else:
    args, kw, flops = (q, k, v), dict(heads=H, scale=0.125, include_self=True, q_prescaled=True), attn_flops(S, L, L, C)


def poll(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "-d", "0"], capture_output=True, text=True, timeout=5).stdout
            pw = re.findall(r"Power \(W\):\s*([\d.]+)", r)
            sc = re.findall(r"sclk clock level:\s*\d+:?\s*\(?(\d+)Mhz", r)
            if pw and sc:
                out.append((float(pw[0]), float(sc[0])))
        except Exception:
            pass
        time.sleep(0.25)


def sample(fn, seconds):
    stop, out = threading.Event(), []
    th = threading.Thread(target=poll, args=(stop, out)); th.start()
    res = fn(seconds)
    stop.set(); th.join()
    body = out[1:] if len(out) > 2 else out
    w = sum(x for x, _ in body) / max(1, len(body))
    clk = sum(c for _, c in body) / max(1, len(body))
    return res, w, clk


def run(var):
    ops.set_attn_variant(var)
    try:
        ops.shared_attention(*args, **kw)
    except Exception as e:
        return None
    torch.cuda.synchronize()

    def body(seconds):
        t0 = time.time(); n = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < seconds:
            for _ in range(50):
                ops.shared_attention(*args, **kw)
            n += 50
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    ms, w, clk = sample(body, secs)
    return ms, w, clk


_, idle_w, idle_clk = sample(lambda s: time.sleep(s), idle_secs)
LADDER = [(20, 63, "matrix skeleton: QK^T and PV MFMAs fed by K / V^T fragment reads from LDS (two resident tiles)"),
          (21, 55, "+ exponentials (64 v_exp_f32 per wave and tile)"),
          (22, 39, "+ row sums (64 adds) and the outgrown-reference check"),
          (23, 7, "+ fp32 -> 16-bit conversions of the probabilities (32 v_cvt_pk)"),
          (24, 3, "+ Q fragments re-read from LDS for every K step (QS form: 8 KiB per wave and tile)"),
          (13, 0, "+ LDS-DMA of every K / V tile and the per-tile barrier = the product kernel")]
LOO = [(25, 8, "exponentials"), (26, 16, "row sums"), (27, 32, "conversions"), (28, 4, "Q re-reads"), (24, 3, "LDS-DMA + barrier")]
print(f"# energy budget of the dominant kernel, shape {shape}: B={B} N={N} L={L} H={H} {'shared (t=1, AdaIN fold)' if shared else 'K/V capture'}, bf16, pre-scaled Q, N(0,1) operands; {secs:.1f} s sustained per row")
print(f"# idle board: {idle_w:.0f} W at {idle_clk:.0f} MHz ({idle_secs:.0f} s, nothing running)")
print("# row | mask | ms/launch | W avg | sclk MHz | J/launch | J above idle | what")
res = {}
for var, mask, what in LADDER + LOO:
    if var in res:
        continue
    r = run(var)
    if r is None:
        print(f"v{var} unavailable (needs the development build: IR_LIB_PATH=tools/experiments/libinstantrestore_hip_dev.so)")
        continue
    ms, w, clk = r
    res[var] = (ms, w, clk, w * ms * 1e-3, (w - idle_w) * ms * 1e-3)
    print(f"v{var:2d} | {mask:2d} | {ms:8.4f} | {w:6.0f} | {clk:5.0f} | {res[var][3]:7.4f} | {res[var][4]:7.4f} | {what}", flush=True)
ops.set_attn_variant(0)
if 13 in res:
    full = res[13]
    print(f"# product kernel: {full[3]:.4f} J per launch = {full[3] / flops * 1e12:.3f} pJ per algorithmic flop, {flops / full[0] / 1e9:.1f} TFLOP/s")
    print("# ladder: increments (they telescope to the product kernel)")
    prev = 0.0
    tot = 0.0
    for var, mask, what in LADDER:
        if var not in res:
            continue
        inc = res[var][3] - prev
        tot += inc
        print(f"#   {inc:+8.4f} J  {100 * inc / full[3]:5.1f} %  (ms {res[var][0]:.4f})  {what}")
        prev = res[var][3]
    print(f"#   sum {tot:.4f} J vs product {full[3]:.4f} J")
    print("# leave-one-out: product kernel minus the class (J saved, share of the launch, ms saved)")
    for var, mask, what in LOO:
        if var in res:
            d = full[3] - res[var][3]
            print(f"#   {d:+8.4f} J  {100 * d / full[3]:5.1f} %  {full[0] - res[var][0]:+.4f} ms   without {what}")
