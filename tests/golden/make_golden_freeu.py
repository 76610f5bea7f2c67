#!/usr/bin/env python3
"""Golden vectors for FreeU.  Run HERE (build container, /root/reference present):

    python -B tests/golden/make_golden_freeu.py

What is pinned and what is not:
* ``apply_freeu`` is the REFERENCE's own function: its ``FunctionDef`` is taken out of
  ``/root/reference/face_replace/models/unet_2d_condition/block.py`` (:3495-3520) with ``ast`` and executed here - the module
  itself cannot be imported (its first lines import ``diffusers``, absent from this image).
* ``fourier_filter`` - the one name that function calls - lives in ``diffusers==0.24.0``
  (``diffusers/utils/torch_utils.py``; imported at block.py:19), third-party code that is NOT under ``/root/reference`` and
  not installed.  Below is a transcription of that release's published body, run on ``torch.fft``.  It is a
  restatement: this seam stays UNPINNED until a real diffusers is available.
The fixture holds inputs' seeds + outputs only (data, no source text).
"""
import ast
import os
import sys

import numpy as np
import torch
from torch.fft import fftn, fftshift, ifftn, ifftshift

sys.dont_write_bytecode = True
REF_BLOCK = "/root/reference/face_replace/models/unet_2d_condition/block.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "freeu_golden.npz")


def fourier_filter(x_in, threshold, scale):
    """diffusers 0.24.0 ``fourier_filter``, transcribed (see the module docstring)."""
    x = x_in
    B, C, H, W = x.shape
    if (W & (W - 1)) != 0 or (H & (H - 1)) != 0:   # non-power-of-2 images must be float32
        x = x.to(dtype=torch.float32)
    x_freq = fftn(x, dim=(-2, -1))
    x_freq = fftshift(x_freq, dim=(-2, -1))
    B, C, H, W = x_freq.shape
    mask = torch.ones((B, C, H, W), device=x.device)
    crow, ccol = H // 2, W // 2
    mask[..., crow - threshold: crow + threshold, ccol - threshold: ccol + threshold] = scale
    x_freq = x_freq * mask
    x_freq = ifftshift(x_freq, dim=(-2, -1))
    x_filtered = ifftn(x_freq, dim=(-2, -1)).real
    return x_filtered.to(dtype=x_in.dtype)


def reference_apply_freeu():
    tree = ast.parse(open(REF_BLOCK).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "apply_freeu")
    mod = ast.Module(body=[fn], type_ignores=[])
    ns = {"torch": torch, "Tuple": tuple.__class_getitem__ if hasattr(tuple, "__class_getitem__") else None,
          "fourier_filter": fourier_filter}
    import typing
    ns["Tuple"] = typing.Tuple
    exec(compile(mod, REF_BLOCK, "exec"), ns)
    return ns["apply_freeu"]


def main():
    apply_freeu = reference_apply_freeu()
    kw = dict(s1=0.9, s2=0.2, b1=1.4, b2=1.6)      # pix2pix_turbo.py:62-68
    cases = []
    out = {}
    shapes = [(2, 8, 16, 16), (1, 6, 32, 32), (1, 4, 12, 20)]
    for ci, shape in enumerate(shapes):
        for res_idx in (0, 1, 2):
            for dt in ("f32", "f16", "bf16"):
                g = torch.Generator().manual_seed(1000 + 10 * ci + res_idx)
                tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
                hidden = torch.randn(shape, generator=g).to(tdt)
                res = (torch.randn(shape, generator=g) * 1.5 + 0.25).to(tdt)
                h_out, r_out = apply_freeu(res_idx, hidden.clone(), res.clone(), **kw)
                cid = "c%d_r%d_%s" % (ci, res_idx, dt)
                cases.append(dict(id=cid, shape=list(shape), res_idx=res_idx, dtype=dt, seed=1000 + 10 * ci + res_idx))
                out[cid + "/hidden_out"] = h_out.float().numpy()
                out[cid + "/res_out"] = r_out.float().numpy()
                out[cid + "/in_checksum"] = np.array([hidden.double().sum().item(), res.double().sum().item()])
    import json
    out["manifest"] = np.frombuffer(json.dumps(dict(cases=cases, kw=kw)).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(cases), "cases")


if __name__ == "__main__":
    main()
