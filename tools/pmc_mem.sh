#!/usr/bin/env bash
# Memory-path PMC passes (counters only): where a kernel's vector memory time goes.
# usage: tools/pmc_mem.sh <tag> <kernel-name-substring> -- <python script and args>
set -u
TAG=$1; KN=$2; shift 3
R=$PWD; OUT=$R/gpurun_out/pmcm_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
# (TA_* / TCP_* counter passes hung rocprofv3 on this pool - 15 GPU-minutes until the limit killed it - and are left out)
P5="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum"
i=0
for P in "$P1" "$P2" "$P5"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT -o pass$i -- python "$@" > $OUT/pass$i.log 2>&1
done
python3 - "$OUT" "$KN" <<'PY'
import csv, glob, sys, collections
out, kn = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-1], {k: round(sum(v) / len(v), 1) for k, v in agg.items()}, "n=%d" % max(len(v) for v in agg.values()) if agg else "")
PY
