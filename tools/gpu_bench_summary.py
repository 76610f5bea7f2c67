"""print the headline, the dominant kernel and the per-class table of a bench.py JSON line.  usage: gpu_bench_summary.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:80], "|", d["value"], d["unit"], d["ms_per_step"], "ms/step | dominant", d["roofline"]["frac"], d["roofline"]["ms_per_launch"], "ms")
kc = d["roofline"].get("step", {}).get("kernel_classes_one_stream", {})
for k, v in kc.items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print("  %-40s %.4f ms  mfma %s  hbm %s" % (k, v["ms_per_step"], v.get("frac_of_mfma_peak"), v.get("frac_of_hbm_peak")))
print("  reconciliation", kc.get("_reconciliation"))
ex = d["config"].get("extras", {})
for k in ("one_stream", "eager_two_streams", "hip_graph", "kv_cached", "cfg1gpu"):
    if k in ex:
        print(" ", k, json.dumps(ex[k])[:300])
