"""`python bench.py --gpus N` launches itself: with N > 1 and no WORLD_SIZE in the environment the
script re-executes as N ranks under torch.distributed.run (rendezvous on 127.0.0.1) - the driver's N = 1 command shape works
for N = 8 without a wrapper - while the torchrun form keeps working.  On this CPU box the GPU step cannot run, so the ranks
take the control-plane-only path (IR_BENCH_CONTROL_ONLY=1): rendezvous over gloo, the barrier / MAX-over-ranks reduction of
the timed region, the scatter / gather leg of SURVEY 8e over the process group, ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd):
    env = dict(os.environ, IR_BENCH_CONTROL_ONLY="1", IR_BENCH_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, timeout=600, env=env)
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_bench_gpus_2_launches_its_own_ranks():
    d, err = _run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert "torch.distributed.run" in err                      # it said what it became
    assert d["control_plane_only"] is True and d["value"] is None
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["global_batch"] == 16
    assert d["config"]["parallelism"].startswith("dp2") and d["scaling"] == "weak"
    assert d["max_over_ranks_s"] >= 0.02                       # rank 1 "stepped" for 20 ms: the MAX, not rank 0's 10 ms
    per = d["config"]["per_rank_ms_per_step"]                 # round 6: every rank's own time beside the MAX (2 steps each)
    assert len(per) == 2 and 4.5 <= per[0] < per[1] and per[1] >= 9.5 and abs(max(per) * 2e-3 - d["max_over_ranks_s"]) < 2e-3
    sg = d["config"]["extras"]["scatter_gather"]
    assert sg["ok"] and sg["rccl_ranks"] == 2 and sg["backend"] == "gloo" and d["config"]["scatter_gather_ms"] == sg["scatter_gather_ms"]


def test_bench_torchrun_form_still_works():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    d, _ = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2
