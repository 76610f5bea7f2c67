"""Host side of the image paths (CPU): the library's host-computed Lanczos tap tables equal the
oracle's restatement of Pillow bit for bit, and the size rules match."""
import numpy as np
import pytest
import torch

from oracle import image_oracle as IO


@pytest.mark.parametrize("n_in,n_out", [(100, 32), (32, 100), (64, 64), (513, 512), (7, 3), (1, 5), (1000, 512),
                                        (4000, 512), (683, 512)])
def test_library_tap_tables_equal_the_oracle(n_in, n_out):
    from instantrestore_amd import ops
    bounds, kk = ops.lanczos_coeffs(n_in, n_out)
    b_ref, k_ref = IO.lanczos_coeffs_np(n_in, n_out)
    assert bounds.dtype == torch.int32 and kk.dtype == torch.int32
    assert np.array_equal(bounds.numpy(), b_ref)
    assert np.array_equal(kk.numpy(), k_ref)


def test_size_rules_match_the_oracle():
    from instantrestore_amd.preprocess import center_crop_offsets, resize_output_size
    for h, w in ((600, 800), (800, 600), (512, 512), (513, 1025), (2000, 3001), (37, 53)):
        for size in (32, 512):
            assert resize_output_size(h, w, size) == IO.resize_output_size(h, w, size)
            oh, ow = resize_output_size(h, w, size)
            assert center_crop_offsets(oh, ow, size) == IO.center_crop_offsets(oh, ow, size)


def test_image_paths_have_no_cpu_fallback():
    from instantrestore_amd import freeu
    from instantrestore_amd.preprocess import LanczosPreprocessor
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LanczosPreprocessor(32)([torch.zeros(40, 40, 3, dtype=torch.uint8)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        freeu.fourier_filter(torch.zeros(1, 2, 8, 8), 1, 0.9)
    with pytest.raises(ValueError):
        LanczosPreprocessor(32)([])
