"""SURVEY.md 8d's "max-abs deviation vs the CPU restatement per dtype" as a TEST (round 6): the product path -
K/V-capturing processor -> harvest -> shared processor, fp32 weights and activations under autocast, projections included - on one
identity against the fp32 torch-CPU port of the reference operator sequence (oracle/shared_attn_oracle.py) on the same
16-bit-representable weights and activations.  One layer of each of the three layer classes of a 512-px identity with 4 references,
AdaIN on, self block included (cfg 2's setting).  The bound is absolute, like north_star's:

    bf16: max|device - port| <= 2.5e-3        fp16: <= 3e-4          (|port| is 0.3 ... 0.5 here)

What the number contains: the 16-bit roundings of q / k / v, of the K/V stash and of the attention output on the device; bench.py
reports the same quantity on its line (`cpu_baseline.deviation_of_device_path`, all nine layers) through the same function."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

BOUND = {torch.bfloat16: 2.5e-3, torch.float16: 3e-4}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("L,C,H", [(256, 1280, 20), (1024, 640, 10), (4096, 320, 5)], ids=["16x16", "32x32", "64x64"])
def test_device_path_deviation_from_the_cpu_port(L, C, H, dtype):
    import bench
    N = 4
    torch.manual_seed(1234 + L)
    ly = bench.baseline_layer(L, C, N, dtype)
    _, ly["port_out"] = bench.baseline_pass(ly, H, N, True, True)
    err, ref_max = bench.device_path_deviation(ly, H, N, True, True, dtype, torch.device("cuda:0"))
    log = os.environ.get("IR_PARITY_LOG")
    if log:
        import json
        with open(log, "a") as f:
            f.write(json.dumps({"what": f"device_path_deviation L={L}", "test": os.environ.get("PYTEST_CURRENT_TEST", ""),
                                "dtype": str(dtype).replace("torch.", ""), "err": err, "ref_max": ref_max, "stated": BOUND[dtype],
                                "regression": BOUND[dtype], "err_over_regression": err / BOUND[dtype]}) + "\n")
    assert 0.05 < ref_max < 5.0, ref_max          # the comparison is not vacuous
    assert err <= BOUND[dtype], f"L={L} {dtype}: device path deviates {err:.3e} from the CPU port (bound {BOUND[dtype]:.1e}, max|port| {ref_max:.3f})"
