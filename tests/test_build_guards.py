"""Build guards that need no GPU (hipcc cross-compiles gfx950 here): the projection GEMM kernels must not touch scratch
inside a loop that issues MFMAs.  A scratch reload there is waited for with s_waitcnt vmcnt(0), which drains the Y stores
and LDS-DMA transfers those loops keep in flight on purpose: the K = 320 X-stationary kernel lost 9 us of 49 to one
(rounds 2-4, found in round 4).  tools/check_resources.py (part of csrc/build.sh) sees totals only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("src", ["linear_skinny.hip", "linear_tiled.hip"])
def test_no_scratch_traffic_inside_the_gemm_main_loops(src):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_loop_scratch.py"),
                        os.path.join(ROOT, "instantrestore_amd", "csrc", src), "--fail", "linear"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
