"""The caller's input transform on the device.

``face_replace/inference/test.py:54-59`` builds, and ``:76,:127,:150`` apply per image on the CPU::

    transforms.Resize(512, interpolation=LANCZOS) -> CenterCrop(512) -> ToTensor() -> Normalize(.5, .5)

:class:`LanczosPreprocessor` does the same for a whole batch of differently sized ``uint8`` RGB
images that already sit in HBM (decoded upstream, or uploaded raw: H*W*3 bytes instead of a
float tensor), in two launches: Pillow's 8-bit horizontal pass over the crop's columns, then the
vertical pass fused with ``/255``, ``(x-0.5)/0.5``, HWC->CHW and the cast.  The resampled bytes are
bit-identical to ``PIL.Image.resize`` (pillow==10.4.0 algorithm); sizes and crop offsets follow
torchvision==0.15.2.  Host work is the tap tables only (``ir_lanczos_coeffs``, cached per
``(in_size, out_size)`` and kept on the device).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Sequence, Tuple

import torch

from . import _lib
from . import ops as _ops


def resize_output_size(in_h: int, in_w: int, size: int) -> Tuple[int, int]:
    """torchvision ``Resize(int)``: short edge -> ``size``, long edge ``int(size * long / short)``."""
    short, long = (in_w, in_h) if in_w <= in_h else (in_h, in_w)
    new_long = int(size * long / short)
    return (new_long, size) if in_w <= in_h else (size, new_long)   # (out_h, out_w)


def center_crop_offsets(h: int, w: int, size: int) -> Tuple[int, int]:
    """torchvision ``CenterCrop``: ``int(round((h - size) / 2.0))`` (round-half-to-even)."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


class LanczosPreprocessor:
    """``pre = LanczosPreprocessor(512, torch.float16); batch = pre(list_of_uint8_HWC_cuda_tensors)``"""

    def __init__(self, size: int = 512, dtype: torch.dtype = torch.float16):
        self.size, self.dtype = int(size), dtype
        self._pitch = (self.size * 3 + 3) // 4 * 4
        self._tables: Dict[Tuple[int, int, torch.device], Tuple[torch.Tensor, ...]] = {}
        self._plans: Dict[Tuple[int, int, torch.device], Tuple[_lib.ImageDesc, int]] = {}
        self._tmp: Dict[torch.device, torch.Tensor] = {}

    def _axis(self, n_in: int, n_out: int, device: torch.device):
        key = (n_in, n_out, device)
        hit = self._tables.get(key)
        if hit is None:
            bounds, kk = _ops.lanczos_coeffs(n_in, n_out)
            # host bounds: row / column range of the crop; device taps in both layouts (the
            # horizontal pass reads them tap-major so neighbouring lanes read neighbouring ints)
            hit = (bounds, bounds.to(device), kk.to(device), kk.t().contiguous().to(device))
            self._tables[key] = hit
        return hit

    def _plan(self, in_h: int, in_w: int, device: torch.device):
        """everything about one source size that does not depend on the pixels: a prototype
        descriptor (sizes, crop, table pointers, row/column ranges) and its scratch bytes"""
        key = (in_h, in_w, device)
        hit = self._plans.get(key)
        if hit is None:
            size = self.size
            out_h, out_w = resize_output_size(in_h, in_w, size)
            top, left = center_crop_offsets(out_h, out_w, size)
            bh_host, bh, _, kh_t = self._axis(in_w, out_w, device)
            bv_host, bv, kv, _ = self._axis(in_h, out_h, device)
            rows, cols = bv_host[top:top + size], bh_host[left:left + size]
            d = _lib.ImageDesc()
            d.in_h, d.in_w, d.out_h, d.out_w = in_h, in_w, out_h, out_w
            d.crop_top, d.crop_left = top, left
            d.bounds_h, d.kk_h, d.bounds_v, d.kk_v = bh.data_ptr(), kh_t.data_ptr(), bv.data_ptr(), kv.data_ptr()
            d.ksize_h, d.ksize_v = kh_t.shape[0], kv.shape[1]
            d.row_first, d.col_first = int(rows[:, 0].min()), int(cols[:, 0].min())
            d.row_count = int((rows[:, 0] + rows[:, 1]).max()) - d.row_first
            d.col_count = int((cols[:, 0] + cols[:, 1]).max()) - d.col_first
            hit = (d, (d.row_count * self._pitch + 255) // 256 * 256)
            self._plans[key] = hit
        return hit

    def prepare(self, images: Sequence[torch.Tensor]):
        """descriptor array for a batch (host work only: no launch)"""
        if len(images) == 0:
            raise ValueError("empty batch")
        device = images[0].device
        descs = (_lib.ImageDesc * len(images))()
        offsets, tmp_bytes = [], 0
        for i, im in enumerate(images):
            if not im.is_cuda:
                raise RuntimeError("LanczosPreprocessor runs on the MI355X only; got a CPU tensor (no CPU fallback)")
            if im.device != device:
                raise RuntimeError("all images must be on one device")
            if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or im.stride(2) != 1 or im.stride(1) != 3:
                raise ValueError("images must be uint8 (H, W, 3) with packed RGB pixels")
            proto, nbytes = self._plan(int(im.shape[0]), int(im.shape[1]), device)
            C.memmove(C.byref(descs[i]), C.byref(proto), C.sizeof(_lib.ImageDesc))
            descs[i].src, descs[i].src_row_bytes = im.data_ptr(), im.stride(0)
            offsets.append(tmp_bytes)
            tmp_bytes += nbytes
        tmp = self._tmp.get(device)
        if tmp is None or tmp.numel() < tmp_bytes:
            tmp = torch.empty(tmp_bytes, dtype=torch.uint8, device=device)
            self._tmp[device] = tmp
        base = tmp.data_ptr()
        for i, off in enumerate(offsets):
            descs[i].tmp = base + off
        return descs, device

    def run(self, descs, device) -> torch.Tensor:
        return _ops.preprocess_lanczos(descs, self.size, self.dtype, device)

    def __call__(self, images: Sequence[torch.Tensor]) -> torch.Tensor:
        return self.run(*self.prepare(images))
