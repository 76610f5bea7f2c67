"""board power and shader clock while ONE kernel runs back to back (rocm-smi sampled from a side thread)
usage: _power_probe.py M N K kernel [fp32]   (env ZERO=1: all-zero operands)"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instantrestore_amd import ops
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kid = ops.LIN_KERNELS[sys.argv[4]]
f32 = len(sys.argv) > 5 and sys.argv[5] == "fp32"
x = torch.randn(M, K, device="cuda", dtype=torch.float32 if f32 else torch.bfloat16)
w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
if os.environ.get("ZERO"):
    x.zero_(); w.zero_()
samples = []
stop = False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip().splitlines())
        except Exception as e:   # noqa
            samples.append([repr(e)])
        time.sleep(0.3)


th = threading.Thread(target=sampler); th.start()
t_end = time.time() + 6.0
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() < t_end:
    for _ in range(200):
        ops.linear(x, w, kernel=kid)
    n += 200
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
print("ZERO" if os.environ.get("ZERO") else "rand", sys.argv[1:], "%.1f us per launch over %d launches" % (e0.elapsed_time(e1) / n * 1e3, n))
for s in samples[2:6]:
    print("   ", " | ".join(s[-2:])[:400])
