"""The image-side oracle (oracle/image_oracle.py) against Pillow's own bytes
(tests/golden/image_golden.npz, made by tests/golden/make_golden_image.py) and against itself
(closed-form FreeU filter vs the literal FFT sequence).  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import image_oracle as IO

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def image_golden():
    z = np.load(os.path.join(HERE, "golden", "image_golden.npz"))
    meta = json.loads(bytes(z["manifest"]).decode())
    return z, meta


def test_lanczos_restatement_matches_pillow_bytes(image_golden):
    z, meta = image_golden
    assert len(meta["cases"]) >= 10
    for m in meta["cases"]:
        img, ref = z[m["id"] + "_in"], z[m["id"] + "_resized"]
        assert IO.resize_output_size(m["in_h"], m["in_w"], m["size"]) == (m["out_h"], m["out_w"])
        got = IO.pil_resize_lanczos_np(img, m["out_w"], m["out_h"])
        assert got.shape == ref.shape and np.array_equal(got, ref), m


def test_coefficients_are_normalised_fixed_point():
    for n_in, n_out in ((100, 32), (32, 100), (64, 64), (513, 512), (7, 3)):
        bounds, kk = IO.lanczos_coeffs_np(n_in, n_out)
        s = kk.sum(axis=1)
        assert np.abs(s - (1 << IO.PRECISION_BITS)).max() <= kk.shape[1]  # rounding of each tap only
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= n_in).all()
    b, k = IO.lanczos_coeffs_np(64, 64)   # scale 1: the centre tap carries everything -> identity
    assert (k.max(axis=1) == 1 << IO.PRECISION_BITS).all()


def test_torchvision_size_rules():
    # Resize(int): short edge -> size, long edge truncated; CenterCrop: banker's rounding
    assert IO.resize_output_size(600, 800, 512) == (512, 682)
    assert IO.resize_output_size(800, 600, 512) == (682, 512)
    assert IO.resize_output_size(512, 512, 512) == (512, 512)
    assert IO.center_crop_offsets(683, 512, 512) == (86, 0)      # 85.5 -> 86 (even)
    assert IO.center_crop_offsets(685, 512, 512) == (86, 0)      # 86.5 -> 86 (even)
    assert IO.center_crop_offsets(682, 512, 512) == (85, 0)


def test_preprocess_is_to_tensor_then_normalize(image_golden):
    z, meta = image_golden
    m = meta["cases"][2]
    t, crop = IO.preprocess_np(z[m["id"] + "_in"], m["size"])
    assert t.dtype == np.float32 and t.shape == (3, m["size"], m["size"]) and crop.shape == (m["size"], m["size"], 3)
    top, left = IO.center_crop_offsets(m["out_h"], m["out_w"], m["size"])
    assert np.array_equal(crop, z[m["id"] + "_resized"][top:top + m["size"], left:left + m["size"]])
    assert t.min() >= -1.0 and t.max() <= 1.0
    lut = (np.arange(256, dtype=np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    assert np.array_equal(t, lut[crop].transpose(2, 0, 1))
    assert lut[0] == -1.0 and lut[255] == 1.0


@pytest.mark.parametrize("shape", [(2, 3, 8, 8), (1, 2, 16, 16), (1, 1, 32, 32), (1, 2, 7, 10), (1, 1, 9, 9)])
@pytest.mark.parametrize("threshold,scale", [(1, 0.9), (1, 0.2), (2, 0.5)])
def test_freeu_closed_form_equals_fft_sequence(shape, threshold, scale):
    rng = np.random.default_rng(3)
    x = rng.standard_normal(shape)
    a = IO.fourier_filter_np(x, threshold, scale)
    b = IO.fourier_filter_closed_form_np(x, threshold, scale)
    assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(a).max())
    assert np.abs(a - x).max() > 1e-3          # the filter does something


def test_apply_freeu_touches_only_the_first_two_resolutions():
    rng = np.random.default_rng(4)
    h, r = rng.standard_normal((1, 4, 8, 8)), rng.standard_normal((1, 4, 8, 8))
    h0, r0 = IO.apply_freeu_np(0, h, r, s1=0.9, s2=0.2, b1=1.4, b2=1.6)
    assert np.allclose(h0[:, :2], h[:, :2] * 1.4) and np.array_equal(h0[:, 2:], h[:, 2:])
    assert np.allclose(r0, IO.fourier_filter_np(r, 1, 0.9))
    h2, r2 = IO.apply_freeu_np(2, h, r, s1=0.9, s2=0.2, b1=1.4, b2=1.6)
    assert np.array_equal(h2, h) and np.array_equal(r2, r)
