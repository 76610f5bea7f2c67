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
                       capture_output=True, text=True, cwd=REPO, timeout=900)
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
