#!/usr/bin/env bash
# kernel timeline of ONE hipGraph replay of the bench step (two streams): start, duration, overlap, gaps
# usage: tools/_step_timeline.sh [extra bench flags]
R=$PWD; OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last replay = the last 1/6 of our kernels... find it as the last run of kernels separated from the previous by the attention pattern: take the final N kernels where N = count per step
names = [r[2] for r in rows]
# count per step: kernels between consecutive occurrences of the first kernel name of the final region; simpler: use gaps > 50 us? steps are back to back. Use count: total kernels in timed steps / steps
import re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:70]
# last step: walk back from the end until we have seen 18 attention w64/pipe "big" launches... use the bench structure: 9 capture + 9 shared attention launches
att = [i for i, n in enumerate(names) if "shared_attn_fwd" in n and "combine" not in n]
per_step = 18 + 0
# remainder-split layers launch two attention kernels; count launches in the last step by locating the 4 timed steps as equal slices of the tail
tail = rows[-(len(rows) // 1):]
# heuristic: timed steps are the last 4 of (2 warm-up eager + capture + 2 replay warm + 4 timed); take the last len/… -> find period by autocorrelation of names
L = len(names)
period = None
for p in range(30, 400):
    if names[L - p:] == names[L - 2 * p:L - p]:
        period = p; break
print("kernels per replay:", period)
step = rows[L - period:]
t0 = step[0][0]
busy_end = t0; gaps = 0
print(f"{'start us':>9} {'dur us':>8} {'gap':>6}  kernel")
for s, e, n in step:
    gap = (s - busy_end) / 1e3
    if gap > 0: gaps += gap
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:6.1f}  {short(n)}")
    busy_end = max(busy_end, e)
tot = (busy_end - t0) / 1e3
print("replay span %.1f us, idle gaps (no kernel running) %.1f us, sum of kernel durations %.1f us" % (tot, gaps, sum(e - s for s, e, _ in step) / 1e3))
PY
