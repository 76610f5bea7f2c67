"""bench.py's output contract: ONE JSON line with the driver's fields plus the roofline and cpu_baseline
objects; without a GPU it refuses to run (there is no CPU path to fall back to)."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_bench_refuses_to_run_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, cwd=REPO, timeout=600)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, cwd=REPO, timeout=900, env=dict(os.environ, IR_BENCH_POWER_SECONDS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "shared_attn" in rf["kernel"]
    pw = d["config"]["extras"]["power"]      # board power / shader clock beside a sustained run: readings, or the reason there are none
    assert "error" in pw or (0 < pw["watts_avg"] <= 1.1 * pw["board_cap_watts"] and 0 < pw["sclk_mhz_avg"] <= pw["sclk_peak_mhz"] + 50)
    if "error" not in pw:
        assert abs(rf["at_measured_clock"]["frac_at_clock"] * rf["at_measured_clock"]["peak_at_clock"] - rf["achieved"]) < 1.0


@pytest.mark.gpu
def test_bench_two_ranks_control_flow_on_one_gpu():
    """The N = 2 launch exactly as the driver issues it (torch.distributed.run, one process per rank), with the
    test hooks that put both ranks on cuda:0 and run the two control-plane collectives over gloo: rank 0 prints the
    one line, n_gpus = 2, the whole-job value counts both ranks' identities over the MAX of the ranks' times."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, IR_BENCH_DIST_BACKEND="gloo", IR_BENCH_SHARE_DEVICE="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline"],
                       capture_output=True, text=True, cwd=REPO, timeout=900, env=env)
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 16
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    assert d["cpu_baseline"] is None and d["config"]["parallelism"].startswith("dp2")


@pytest.mark.gpu
def test_bench_line_survives_a_rank_that_stalls_in_the_collective_extra():
    """The scatter/gather extra runs after the headline measurement and under a watchdog when N > 1: if a rank stalls
    inside it (test hook: rank 1 never enters), rank 0 still prints its one line - with the error recorded where the
    extra's result would be - and every rank leaves with exit code 0 instead of waiting in a collective for ever."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, IR_BENCH_DIST_BACKEND="gloo", IR_BENCH_SHARE_DEVICE="1", IR_BENCH_SG_HANG="1", IR_BENCH_SG_TIMEOUT="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-roofline"],
                       capture_output=True, text=True, cwd=REPO, timeout=900, env=env)
    assert r.returncode == 0, (r.stderr + r.stdout)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert "error" in d["config"]["extras"]["scatter_gather"]
