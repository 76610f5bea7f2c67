"""ORACLE - test infrastructure only, never the product path.

CPU restatement (numpy) of the data formats either side of the hot path (SURVEY.md section 8f
ranks 3 and 4):

* the caller's input transform, ``face_replace/inference/test.py:54-59``::

      Resize(512, LANCZOS) -> CenterCrop(512) -> ToTensor() -> Normalize(0.5, 0.5)

  on PIL ``RGB`` images (``test.py:177-182``).  The arithmetic lives in two third-party packages
  that are not under ``/root/reference``: ``pillow==10.4.0`` (``environment_new.yml:237``;
  ``src/libImaging/Resample.c``: 8-bit two-pass resampling with 22-bit fixed-point
  coefficients) and ``torchvision==0.15.2`` (``environment_new.yml:331``; output-size rule of
  ``Resize``, the banker's-rounding crop offsets of ``CenterCrop``, ``ToTensor`` = ``/255``,
  ``Normalize`` = ``(x - 0.5) / 0.5`` in float32).  Their published algorithms are restated
  here.  Pinning: Pillow *is* installed in the build container (12.2.0 - the 8-bit resampler is
  unchanged since 4.x), so ``pil_resize_lanczos_np`` is checked byte-for-byte against
  ``PIL.Image.resize`` itself (``tests/golden/make_golden_image.py`` -> ``tests/golden/
  image_golden.npz``, ``tests/test_image_oracle.py``).  torchvision is absent: its four steps are
  restated from the release's source and are "parity unpinned" at that seam.

* FreeU's skip-feature filter, ``face_replace/models/unet_2d_condition/block.py:3495-3520``
  (``apply_freeu`` -> ``diffusers.utils.torch_utils.fourier_filter``, diffusers==0.24.0, third
  party, absent: restated from its published algorithm - "parity unpinned" at that seam; the
  restatement is the literal FFT sequence, in float64).

Who may import this file: ``tests/``, ``__graft_entry__.smoke()``, ``bench.py``'s cpu_baseline.
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: coefficients are ints scaled by 2**22
LANCZOS_SUPPORT = 3.0


# ------------------------------------------------------------------------------------------
# Pillow: Resample.c
# ------------------------------------------------------------------------------------------
def _sinc(x: float) -> float:
    if x == 0.0:
        return 1.0
    x = x * math.pi
    return math.sin(x) / x


def _lanczos(x: float) -> float:
    # truncated sinc, Resample.c lanczos_filter: [-3, 3)
    if -3.0 <= x < 3.0:
        return _sinc(x) * _sinc(x / 3.0)
    return 0.0


def lanczos_coeffs_np(in_size: int, out_size: int):
    """``precompute_coeffs`` + ``normalize_coeffs_8bpc`` of Resample.c for the full-image box.

    Returns ``(bounds (out, 2) int32 = [first source index, tap count], kk (out, ksize) int32)``.
    """
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size   # box is float in C
    filterscale = max(scale, 1.0)
    support = LANCZOS_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_lanczos((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            # C: (int)(+-0.5 + k * (1 << PRECISION_BITS)): truncation toward zero
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis0(img: np.ndarray, bounds: np.ndarray, kk: np.ndarray) -> np.ndarray:
    """one 8-bit pass along axis 0 (the other pass is this on the transposed image)"""
    out = np.empty((bounds.shape[0],) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for yy in range(bounds.shape[0]):
        ymin, ymax = int(bounds[yy, 0]), int(bounds[yy, 1])
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        acc += np.tensordot(kk[yy, :ymax].astype(np.int64), src[ymin:ymin + ymax], axes=(0, 0))
        out[yy] = _clip8(acc)
    return out


def pil_resize_lanczos_np(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """``Image.resize((out_w, out_h), Image.LANCZOS)`` on an (H, W, 3) uint8 image: horizontal
    pass first (rounded to uint8), then vertical (ImagingResampleInner)."""
    in_h, in_w, _ = img.shape
    if (in_w, in_h) == (out_w, out_h):
        return img.copy()                                     # Image.resize early exit
    cur = img
    if out_w != in_w:
        bh, kh = lanczos_coeffs_np(in_w, out_w)
        cur = np.ascontiguousarray(_resample_axis0(np.ascontiguousarray(cur.transpose(1, 0, 2)), bh, kh).transpose(1, 0, 2))
    if out_h != in_h:
        bv, kv = lanczos_coeffs_np(in_h, out_h)
        cur = _resample_axis0(cur, bv, kv)
    return cur


# ------------------------------------------------------------------------------------------
# torchvision 0.15.2 transforms used at test.py:54-59
# ------------------------------------------------------------------------------------------
def resize_output_size(in_h: int, in_w: int, size: int):
    """``Resize(int)``: the short edge becomes ``size``, the long edge ``int(size * long / short)``."""
    short, long = (in_w, in_h) if in_w <= in_h else (in_h, in_w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if in_w <= in_h else (new_short, new_long)   # (out_h, out_w)


def center_crop_offsets(h: int, w: int, size: int):
    """``CenterCrop``: ``int(round((h - size) / 2.0))`` with Python's round-half-to-even."""
    return int(round((h - size) / 2.0)), int(round((w - size) / 2.0))


def preprocess_np(img: np.ndarray, size: int = 512):
    """(H, W, 3) uint8 -> ((3, size, size) float32 in [-1, 1], the uint8 crop it came from)."""
    out_h, out_w = resize_output_size(img.shape[0], img.shape[1], size)
    r = pil_resize_lanczos_np(img, out_w, out_h)
    top, left = center_crop_offsets(out_h, out_w, size)
    crop = r[top:top + size, left:left + size]
    t = crop.transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)        # ToTensor
    t = (t - np.float32(0.5)) / np.float32(0.5)                               # Normalize
    return t, crop


# ------------------------------------------------------------------------------------------
# FreeU (block.py:3495-3520 -> diffusers fourier_filter)
# ------------------------------------------------------------------------------------------
def fourier_filter_np(x: np.ndarray, threshold: int, scale: float) -> np.ndarray:
    """(B, C, H, W) -> same shape, float64: FFT, shift, scale the (2*threshold)^2 centre bins,
    unshift, inverse FFT, real part."""
    x = np.asarray(x, np.float64)
    f = np.fft.fftshift(np.fft.fftn(x, axes=(-2, -1)), axes=(-2, -1))
    h, w = x.shape[-2:]
    mask = np.ones(x.shape, np.float64)
    crow, ccol = h // 2, w // 2
    mask[..., crow - threshold:crow + threshold, ccol - threshold:ccol + threshold] = scale
    f = np.fft.ifftshift(f * mask, axes=(-2, -1))
    return np.fft.ifftn(f, axes=(-2, -1)).real


def apply_freeu_np(resolution_idx, hidden, res_hidden, s1, s2, b1, b2):
    """``apply_freeu`` (block.py:3495-3520): backbone half-channel gain + skip-feature filter for
    the first two up-block resolutions; returns float64 arrays."""
    hidden = np.array(hidden, np.float64)
    res_hidden = np.array(res_hidden, np.float64)
    if resolution_idx in (0, 1):
        b, s = (b1, s1) if resolution_idx == 0 else (b2, s2)
        half = hidden.shape[1] // 2
        hidden[:, :half] = hidden[:, :half] * b
        res_hidden = fourier_filter_np(res_hidden, 1, s)
    return hidden, res_hidden


def fourier_filter_closed_form_np(x: np.ndarray, threshold: int, scale: float) -> np.ndarray:
    """Same map without an FFT: only the bins u, v in [-threshold, threshold-1] are touched, so
    ``y = x + (scale-1)/(H*W) * Re sum_{u,v} X(u,v) e^{+2 pi i (u r/H + v c/W)}`` with the
    (2*threshold)^2 DFT coefficients ``X(u,v)`` - the formulation the HIP kernel uses (one read,
    one write).  Checked against ``fourier_filter_np`` in tests/test_image_oracle.py."""
    x = np.asarray(x, np.float64)
    h, w = x.shape[-2:]
    r = np.arange(h)[:, None]
    c = np.arange(w)[None, :]
    y = x.copy()
    for u in range(-threshold, threshold):
        for v in range(-threshold, threshold):
            theta = 2.0 * np.pi * (((u * r) % h) / h + ((v * c) % w) / w)
            co, si = np.cos(theta), np.sin(theta)
            re = (x * co).sum(axis=(-2, -1), keepdims=True)
            im = -(x * si).sum(axis=(-2, -1), keepdims=True)
            y += (scale - 1.0) / (h * w) * (re * co - im * si)
    return y
