#!/usr/bin/env bash
# kernel timeline of ONE hipGraph replay of the bench step (two streams): start, duration, overlap, gaps
# usage: tools/_step_timeline.sh [extra bench flags]
R=$PWD; OUT=$R/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-extras "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(f)))
# the replays (2 warm + 4 timed) are the longest run of kernels without a gap above 200 us
runs, cur, busy = [], [0], rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] - busy > 200000:
        runs.append(cur); cur = []
    cur.append(i); busy = max(busy, rows[i][1])
runs.append(cur)
blk = max(runs, key=len)
per = len(blk) // 6
step = [rows[i] for i in blk[-per:]]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n).replace("void ", "")
    return n[:62]
t0 = step[0][0]; busy = t0; gaps = 0.0
print("kernels per replay: %d" % per)
print("%9s %8s %6s  queue kernel" % ("start us", "dur us", "idle"))
for s, e, n, q in step:
    gap = max((s - busy) / 1e3, 0.0); gaps += gap
    print("%9.1f %8.1f %6.1f  q%s %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, short(n)))
    busy = max(busy, e)
print("replay span %.1f us, no kernel running for %.1f us, sum of kernel durations %.1f us" % ((busy - t0) / 1e3, gaps, sum(e - s for s, e, _, _ in step) / 1e3))
PY
