#!/usr/bin/env bash
# The end-of-change run on the GPU box (one gpurun call): whole -m gpu suite, the bench line, kernel statistics of the
# one-stream step under rocprofv3, smoke().  usage: gpurun -- 'bash tools/gpu_round_check.sh [tag]'; results under gpurun_out/<tag>/
T=${1:-check}
mkdir -p gpurun_out/$T
R=$PWD
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/$T/gputest.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/prof_bench -o b -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-extras --two-streams 0 > $R/gpurun_out/$T/prof_bench.log 2>&1 )
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$T/smoke.txt 2>&1
cat gpurun_out/$T/gputest.txt; tail -2 gpurun_out/$T/smoke.txt; T=$T python - <<'PY'
import json, os
d=json.loads(open("gpurun_out/" + os.environ["T"] + "/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["back_to_back"], d["config"]["extras"]["one_stream"], d["config"]["extras"]["hip_graph"]["images_per_s"])
PY
head -12 gpurun_out/$T/prof_bench/b_kernel_stats.csv | cut -c1-150
