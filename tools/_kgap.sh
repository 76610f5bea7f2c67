#!/usr/bin/env bash
# kernel-trace of back-to-back launches of one GEMM shape: per-dispatch duration and the gap to the next dispatch
# usage: tools/_kgap.sh M N K kernel [fp32]
R=$PWD; OUT=$R/gpurun_out/kgap; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o kg -- python $R/tools/prof_linear.py "$@" > $OUT/log.txt 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "linear" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for a, b in zip(rows, rows[1:] + [None]):
    d = (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3
    g = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 if b else float("nan")
    print(a["Kernel_Name"][:60], "dur %.1f us  gap-to-next %.1f us" % (d, g))
PY
