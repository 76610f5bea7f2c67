"""FreeU against vectors produced by the REFERENCE's own ``apply_freeu`` (block.py:3495-3520, executed from the reference
tree by tests/golden/make_golden_freeu.py) around a transcription of diffusers 0.24.0's ``fourier_filter`` - the package is
absent from this image, so that one callee is restated and the seam stays unpinned.
CPU: the numpy oracle (oracle/image_oracle.py) reproduces the vectors.  GPU: ``instantrestore_amd.freeu.apply_freeu`` does."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "freeu_golden.npz")
Z = np.load(GOLD)
MAN = json.loads(bytes(Z["manifest"]).decode())
DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}
ULP = {"f32": 2.0 ** -20, "f16": 2.0 ** -10, "bf16": 2.0 ** -7}    # one unit in the last place at 1.0 (fp32: the FFT's own noise)


def _inputs(c):
    g = torch.Generator().manual_seed(c["seed"])
    hidden = torch.randn(c["shape"], generator=g).to(DT[c["dtype"]])
    res = (torch.randn(c["shape"], generator=g) * 1.5 + 0.25).to(DT[c["dtype"]])
    chk = Z[c["id"] + "/in_checksum"]
    assert abs(hidden.double().sum().item() - chk[0]) < 1e-9 and abs(res.double().sum().item() - chk[1]) < 1e-9, "RNG drifted"
    return hidden, res


@pytest.mark.parametrize("c", MAN["cases"], ids=[c["id"] for c in MAN["cases"]])
def test_oracle_reproduces_the_reference_apply_freeu(c):
    hidden, res = _inputs(c)
    h, r = IO.apply_freeu_np(c["res_idx"], hidden.float().numpy(), res.float().numpy(), **MAN["kw"])
    # the reference multiplies the backbone half IN the tensor's dtype and rounds the filtered skip to it
    want_h, want_r = Z[c["id"] + "/hidden_out"].astype(np.float64), Z[c["id"] + "/res_out"].astype(np.float64)
    tol = ULP[c["dtype"]]
    assert np.abs(h - want_h).max() <= tol * max(1.0, np.abs(want_h).max())
    assert np.abs(r - want_r).max() <= tol * max(1.0, np.abs(want_r).max())


@pytest.mark.gpu
@pytest.mark.parametrize("c", MAN["cases"], ids=[c["id"] for c in MAN["cases"]])
def test_hip_apply_freeu_reproduces_the_reference(c):
    from instantrestore_amd import freeu
    hidden, res = _inputs(c)
    h, r = freeu.apply_freeu(c["res_idx"], hidden.cuda(), res.cuda(), **MAN["kw"])
    assert h.dtype == DT[c["dtype"]] and r.dtype == DT[c["dtype"]]
    want_h, want_r = Z[c["id"] + "/hidden_out"].astype(np.float64), Z[c["id"] + "/res_out"].astype(np.float64)
    tol = ULP[c["dtype"]]
    assert np.abs(h.double().cpu().numpy() - want_h).max() <= tol * max(1.0, np.abs(want_h).max())
    assert np.abs(r.double().cpu().numpy() - want_r).max() <= tol * max(1.0, np.abs(want_r).max())
