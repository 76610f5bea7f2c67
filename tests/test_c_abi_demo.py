"""The C ABI from plain C (examples/c_abi_demo.c): compiles and links with gcc against the shared library
and the HIP runtime - no torch, no Python types in the signatures - and, on the GPU, reproduces a float64
loop written in C."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_demo")
    cmd = ["gcc", "-std=gnu11", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(REPO, "include"),
           os.path.join(REPO, "examples", "c_abi_demo.c"), "-L" + os.path.join(REPO, "instantrestore_amd"),
           "-linstantrestore_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(REPO, "instantrestore_amd"), "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_demo_compiles_and_links_as_plain_c(tmp_path):
    exe = _build(tmp_path)
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_c_demo_runs_and_matches_its_float64_loop(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
