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


def test_loop_scratch_detector_on_a_synthetic_listing():
    """the guard's own logic, without a compiler: scratch inside an annotated MFMA loop is found, scratch in a cold block that
    merely sits between the loop's labels in the layout - or in a loop without MFMAs - is not"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_loop_scratch as C
    listing = """
_Z3foov: ; @foo
\tscratch_store_dword off, v1, off ; prologue spill: not in a loop
.LBB0_1:                                ; =>This Inner Loop Header: Depth=1
\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]
\ts_cbranch_scc1 .LBB0_3
.LBB0_2:                                ;   in Loop: Header=BB0_1 Depth=1
\tscratch_load_dword v1, off, off ; 4-byte Folded Reload
\ts_branch .LBB0_1
.LBB0_3:
\tscratch_load_dword v2, off, off ; cold block after the loop
.LBB0_4:                                ; =>This Inner Loop Header: Depth=1
\tscratch_load_dword v3, off, off ; a loop without MFMAs
\ts_cbranch_scc1 .LBB0_4
.Lfunc_end0:
""".split("\n")
    (name, s, e), = list(C.kernels(listing))
    found = C.loops_with_scratch(listing, s, e)
    assert name == "_Z3foov" and len(found) == 1
    (_, label), (n_mfma, scr) = next(iter(found.items()))
    assert label == ".LBB0_1" and n_mfma == 1 and len(scr) == 1 and "v1" in scr[0]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_w128_kernel_keeps_the_compiler_out_of_the_accumulator_registers(tmp_path):
    """shared_attn_fwd_w128.hip keeps its state between asm statements in accumulator registers the compiler does not know about
    (O, the Q fragments, m / l / l_done): the ISA must show NO compiler-generated v_accvgpr_* / AGPR operand outside the asm
    blocks, no scratch, no spill, 256 AGPRs and one wave per SIMD - otherwise a compiler copy into that range corrupts the state
    silently (the guide's item 4 of 'what hipcc does not do for an asm statement')."""
    import re
    out = tmp_path / "w128.s"
    src = os.path.join(ROOT, "instantrestore_amd", "csrc", "shared_attn_fwd_w128.hip")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out), src],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    inasm, cur, bad, kernels = False, None, [], set()
    for line in open(out):
        m = re.match(r"^(_ZN\S*shared_attn_fwd_w128_kernel\S*):", line)
        if m:
            cur = m.group(1)
            kernels.add(cur)
        if ";;#ASMSTART" in line:
            inasm = True
            continue
        if ";;#ASMEND" in line:
            inasm = False
            continue
        code = line.split(";")[0]
        if cur and not inasm and (re.search(r"v_accvgpr|\ba\d+\b|a\[\d+", code) or "scratch_" in code):
            bad.append((cur[-30:], line.strip()))
    text = open(out).read()
    assert len(kernels) == 4, kernels                                  # bf16 / f16 x fold / plain
    assert not bad, bad[:10]
    assert text.count(".vgpr_spill_count: 0") == 4 and text.count(".agpr_count:     256") == 4
    assert text.count(".private_segment_fixed_size: 0") >= 4
