import sys, json, io, contextlib
sys.path.insert(0, "/root/repo")
import bench
from instantrestore_amd import attn_processors as ap
res = {}
for rnd in range(3):
    for name, fc, pq in (("fused+presc", True, True), ("nofuse+presc", False, True), ("fused+plain", True, False), ("nofuse+plain", False, False)):
        ap.FUSED_CAST, ap.PRESCALE_Q = fc, pq
        sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extras", "--no-roofline"]
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main()
        d = json.loads(buf.getvalue().strip().splitlines()[-1])
        res.setdefault(name, []).append(d["ms_per_step"])
for k, v in res.items():
    print(k, v)
