# PMC passes for the 32x32-token class (cfg 2 shape): variant 11 (32-row kernel) and 16 (128-row kernel)
R=$PWD; O=$R/gpurun_out/r6n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
for V in 11 16; do
  for i in 1 3; do
    eval P=\$P$i
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_v$V -o pass$i -- python $R/tools/prof_attn.py $V 3 1024 10 1 1 1 > $O/pmc_v${V}_pass$i.log 2>&1
  done
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/pmc_v*/*counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "shared_attn_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-2], f.split("/")[-1], {k: round(sum(v) / len(v), 1) for k, v in agg.items()})
PY
