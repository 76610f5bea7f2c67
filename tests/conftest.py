import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden", "instantrestore_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def bits_to_f32(u16: np.ndarray, lowp: str) -> np.ndarray:
    """raw 16-bit patterns (fp16 / bf16) -> float32 values"""
    u16 = np.asarray(u16, dtype=np.uint16)
    if lowp == "f16":
        return u16.view(np.float16).astype(np.float32)
    return (u16.astype(np.uint32) << 16).view(np.float32)


class Golden:
    def __init__(self):
        z = np.load(GOLDEN)
        self.z = z
        self.manifest = json.loads(bytes(z["manifest"]).decode())

    def cases(self, kind=None):
        return [m for m in self.manifest if kind is None or m["kind"] == kind]

    def arr(self, meta, name, as_f32=True):
        key = f"{meta['id']}/{name}"
        if key not in self.z.files:
            return None
        a = self.z[key]
        if a.dtype == np.uint16 and as_f32:
            return bits_to_f32(a, meta["lowp"])
        return a


@pytest.fixture(scope="session")
def golden():
    return Golden()


def _load_manifest():
    z = np.load(GOLDEN)
    return json.loads(bytes(z["manifest"]).decode())


GOLDEN_MANIFEST = _load_manifest()


def pytest_sessionstart(session):
    """The HIP library is a build artefact (git-ignored).  If the suite is started on a fresh
    checkout, build it first - hipcc cross-compiles for gfx950 without a GPU, ~5 s."""
    import shutil
    import subprocess
    lib = os.path.join(REPO, "instantrestore_amd", "libinstantrestore_hip.so")
    if os.path.exists(lib):
        return
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which("hipcc")):
        return  # the C-ABI tests will say so loudly
    subprocess.run(["bash", os.path.join(REPO, "instantrestore_amd", "csrc", "build.sh")], check=False,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
