#!/usr/bin/env bash
# board power / shader clock while bench.py's step runs (hipGraph replay), sampled from the shell
python bench.py --steps 1500 --warmup 5 > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
sleep 45
for i in 1 2 3 4 5 6 7 8; do rocm-smi --showpower --showclocks --csv | tail -1 | awk -F, '{print "sclk " $7 " power " $NF " W"}'; sleep 0.7; done
wait $BP
python - <<'PY'
import json
d = json.loads(open("gpurun_out/power_bench.json").read().strip().splitlines()[-1])
print("bench:", d["value"], d["unit"], d["ms_per_step"], "ms/step", "frac", d["roofline"]["frac"])
PY
