"""GPU parity of the image-side kernels (through the C ABI): bit-exact for the integer resampler
(against Pillow's own bytes in tests/golden/image_golden.npz and against the oracle on seeded
inputs at full size), stated fp tolerance for the FreeU filter."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def image_golden():
    z = np.load(os.path.join(HERE, "golden", "image_golden.npz"))
    return z, json.loads(bytes(z["manifest"]).decode())


def _lut(dtype):
    f32 = (np.arange(256, dtype=np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)
    return torch.from_numpy(f32).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
def test_preprocess_matches_pillow_bytes(image_golden, dtype):
    from instantrestore_amd.preprocess import LanczosPreprocessor
    z, meta = image_golden
    by_size = {}
    for m in meta["cases"]:
        by_size.setdefault(m["size"], []).append(m)
    lut = _lut(dtype)
    for size, cases in by_size.items():
        pre = LanczosPreprocessor(size, dtype)
        imgs = [torch.from_numpy(z[m["id"] + "_in"]).cuda() for m in cases]
        out = pre(imgs).cpu()                                    # one launch pair for the ragged batch
        assert out.shape == (len(cases), 3, size, size) and out.dtype == dtype
        for i, m in enumerate(cases):
            top, left = IO.center_crop_offsets(m["out_h"], m["out_w"], size)
            crop = z[m["id"] + "_resized"][top:top + size, left:left + size]       # Pillow's bytes
            want = lut[torch.from_numpy(crop.astype(np.int64))].permute(2, 0, 1)
            assert torch.equal(out[i], want), (m, dtype)


def test_preprocess_full_size_batch_equals_oracle():
    """512-px crops from seeded images larger and smaller than the target (down- and up-sampling,
    landscape and portrait, odd sizes, a strided source view, 18 images = two launch chunks)."""
    from instantrestore_amd.preprocess import LanczosPreprocessor
    rng = np.random.default_rng(7)
    shapes = [(600, 800), (1024, 1024), (700, 513), (512, 512), (400, 300), (1201, 900)] * 3
    imgs = []
    for i, (h, w) in enumerate(shapes):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        a[..., i % 3] = ((np.sin(yy / (3.0 + i)) * np.cos(xx / 11.0)) * 127 + 128).clip(0, 255).astype(np.uint8)
        imgs.append(a)
    dev = []
    for i, a in enumerate(imgs):
        if i == 1:   # rows padded: src_row_bytes != 3 * in_w
            buf = torch.zeros((a.shape[0], a.shape[1] + 5, 3), dtype=torch.uint8, device="cuda")
            buf[:, :a.shape[1]] = torch.from_numpy(a).cuda()
            dev.append(buf[:, :a.shape[1]])
        else:
            dev.append(torch.from_numpy(a).cuda())
    out = LanczosPreprocessor(512, torch.float32)(dev).cpu().numpy()
    for i in range(len(imgs)):
        want, _ = IO.preprocess_np(imgs[i], 512)
        assert np.array_equal(out[i], want), (i, shapes[i])
    # idempotence property at scale 1: a 512x512 source is only normalised
    lut = _lut(torch.float32).numpy()
    assert np.array_equal(out[3], lut[imgs[3]].transpose(2, 0, 1))


def test_preprocess_very_wide_source_uses_the_one_row_path():
    """a panorama whose span under the crop (> 5000 columns) does not fit four staged rows in LDS"""
    from instantrestore_amd.preprocess import LanczosPreprocessor
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, (600, 9000, 3), dtype=np.uint8)
    out = LanczosPreprocessor(512, torch.float32)([torch.from_numpy(a).cuda()]).cpu().numpy()
    want, _ = IO.preprocess_np(a, 512)
    assert np.array_equal(out[0], want)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
@pytest.mark.parametrize("shape,thr,scale", [((2, 1280, 8, 8), 1, 0.9), ((2, 1280, 16, 16), 1, 0.2), ((1, 640, 32, 32), 1, 0.2),
                                             ((1, 7, 64, 64), 1, 0.9), ((3, 5, 7, 10), 1, 0.5), ((1, 3, 16, 16), 2, 0.3),
                                             ((1, 2, 9, 9), 3, 1.7)])
def test_freeu_fourier_filter(dtype, shape, thr, scale):
    """tolerance (floating point): fp32 arithmetic on 16-bit inputs, result rounded once to the
    tensor dtype - |err| <= 2e-6*max|x| (fp32) / half an ulp of the output + that (16-bit)."""
    from instantrestore_amd import freeu
    torch.manual_seed(11)
    x = (torch.randn(shape) * 1.5 + 0.3).to(dtype)
    ref = IO.fourier_filter_np(x.float().numpy(), thr, scale)
    y = freeu.fourier_filter(x.cuda(), thr, scale)
    assert y.dtype == dtype and y.shape == x.shape
    err = np.abs(y.float().cpu().numpy().astype(np.float64) - ref).max()
    amax = max(1.0, np.abs(ref).max())
    tol = {torch.float32: 4e-6, torch.float16: 2.0 ** -11 + 4e-6, torch.bfloat16: 2.0 ** -8 + 4e-6}[dtype] * amax
    assert err <= tol, (err, tol)
    # in place, strided channel slice
    big = torch.zeros((shape[0], shape[1] + 2, shape[2], shape[3]), dtype=dtype, device="cuda")
    big[:, 1:-1] = x.cuda()
    y2 = freeu.fourier_filter(big[:, 1:-1], thr, scale)
    assert torch.equal(y2, y)


def test_apply_freeu_contract():
    from instantrestore_amd import freeu
    torch.manual_seed(2)
    h = torch.randn(2, 8, 8, 8, device="cuda", dtype=torch.float16)
    r = torch.randn(2, 8, 8, 8, device="cuda", dtype=torch.float16)
    h0 = h.clone()
    kw = dict(s1=0.9, s2=0.2, b1=1.4, b2=1.6)
    h1, r1 = freeu.apply_freeu(0, h, r, **kw)
    assert h1 is h and torch.equal(h[:, :4], h0[:, :4] * 1.4) and torch.equal(h[:, 4:], h0[:, 4:])
    want_h, want_r = IO.apply_freeu_np(0, h0.float().cpu().numpy(), r.float().cpu().numpy(), **kw)
    assert np.abs(r1.float().cpu().numpy() - want_r).max() <= 2.0 ** -10 * max(1.0, np.abs(want_r).max())
    h2, r2 = freeu.apply_freeu(2, h, r, **kw)
    assert h2 is h and r2 is r
    with pytest.raises(Exception):
        freeu.fourier_filter(torch.zeros(1, 1, 128, 128, device="cuda"), 1, 0.5)   # > 4096 elements per plane


def test_preprocess_properties_constant_and_saturated_images():
    """size-independent properties: a constant image stays constant through any resize (the taps sum to
    2^22 +- rounding and the result is clipped), black -> -1, white -> +1"""
    from instantrestore_amd.preprocess import LanczosPreprocessor
    pre = LanczosPreprocessor(512, torch.float32)
    imgs = []
    for val, (h, w) in ((0, (700, 900)), (255, (333, 512)), (128, (1500, 1100)), (37, (512, 512))):
        imgs.append(torch.full((h, w, 3), val, dtype=torch.uint8, device="cuda"))
    out = pre(imgs).cpu()
    lut = (torch.arange(256, dtype=torch.float32) / 255.0 - 0.5) / 0.5
    for i, val in enumerate((0, 255, 128, 37)):
        assert torch.equal(out[i], torch.full((3, 512, 512), float(lut[val]))), val
    assert float(out[0].max()) == -1.0 and float(out[1].min()) == 1.0


def test_sharded_image_path_on_one_rank_equals_the_unfused_bytes():
    """sharding.run_sharded_images (preprocess output = send buffer, tensor2im before the gather) on ONE rank: the same
    bytes as calling the two kernels around the step by hand; ragged source sizes"""
    from instantrestore_amd import ops, sharding
    from instantrestore_amd.preprocess import LanczosPreprocessor
    g = torch.Generator().manual_seed(2)
    total, n_refs, S = 3, 2, 64
    images = [[torch.randint(0, 256, (70 + 9 * i + j, 90 + 5 * j, 3), generator=g, dtype=torch.uint8).cuda() for j in range(1 + n_refs)]
              for i in range(total)]
    step = lambda d, r: (d.float() * 0.6 + r.float().mean(dim=1) * 0.4).to(torch.float16)
    got = sharding.run_sharded_images(step, images, total, n_refs, S, torch.float16, torch.device("cuda"))
    packed = LanczosPreprocessor(S, torch.float16)([im for ident in images for im in ident]).view(total, 1 + n_refs, 3, S, S)
    want = ops.tensor2im_u8(step(packed[:, 0], packed[:, 1:]))
    assert got.dtype == torch.uint8 and tuple(got.shape) == (total, S, S, 3) and torch.equal(got, want)


def test_randomised_source_sizes_equal_the_oracle():
    """seeded sweep (IR_SWEEP_CASES / IR_SWEEP_SEED widen it): sources from 17 to 1700 pixels a side - far smaller and far larger than
    the target, extreme aspect ratios, odd sizes, padded rows - and targets of 64 ... 512 pixels, one launch per target size over a
    ragged batch: the device bytes must be the oracle's (Pillow's fixed-point Lanczos, resize-shorter-side + centre crop +
    normalise; inference/test.py:54-59) everywhere"""
    from instantrestore_amd.preprocess import LanczosPreprocessor
    seed = int(os.environ.get("IR_SWEEP_SEED", "17"))
    rng = np.random.default_rng(seed)
    cases = int(os.environ.get("IR_SWEEP_CASES", "24"))
    done = 0
    while done < cases:
        size = int(rng.choice([64, 96, 128, 256, 512]))
        n = int(rng.integers(1, 7))
        imgs, dev = [], []
        for i in range(n):
            kind = int(rng.integers(0, 4))
            if kind == 0:      # smaller than the target (up-sampling), down to a sliver
                h, w = int(rng.integers(17, size + 1)), int(rng.integers(17, size + 1))
            elif kind == 1:    # extreme aspect ratio
                h, w = int(rng.integers(17, 200)), int(rng.integers(600, 1700))
                if rng.integers(0, 2):
                    h, w = w, h
            else:
                h, w = int(rng.integers(size // 2, 1700)), int(rng.integers(size // 2, 1700))
            a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            if rng.integers(0, 2):   # smooth content in one channel: long runs of equal taps products, saturation at both ends
                yy, xx = np.mgrid[0:h, 0:w]
                a[..., i % 3] = ((np.sin(yy / 5.0) * np.cos(xx / 9.0)) * 140 + 128).clip(0, 255).astype(np.uint8)
            imgs.append(a)
            if rng.integers(0, 3) == 0:   # rows padded: src_row_bytes != 3 * in_w
                pad = int(rng.integers(1, 9))
                buf = torch.zeros((h, w + pad, 3), dtype=torch.uint8, device="cuda")
                buf[:, :w] = torch.from_numpy(a).cuda()
                dev.append(buf[:, :w])
            else:
                dev.append(torch.from_numpy(a).cuda())
        out = LanczosPreprocessor(size, torch.float32)(dev).cpu().numpy()
        for i, a in enumerate(imgs):
            want, _ = IO.preprocess_np(a, size)
            assert np.array_equal(out[i], want), (seed, done, i, a.shape, size)
        done += n


def test_randomised_freeu_filter_shapes():
    """seeded sweep (IR_SWEEP_CASES / IR_SWEEP_SEED widen it): random plane counts and plane sizes up to the kernel's 4096 elements
    (odd, non-square, 2 x 2), every admissible threshold, both signs of the scale - against the float64 FFT sequence
    (block.py:3495-3520's fourier_filter) at the tolerance of test_freeu_fourier_filter"""
    from instantrestore_amd import freeu
    seed = int(os.environ.get("IR_SWEEP_SEED", "31"))
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    for case in range(int(os.environ.get("IR_SWEEP_CASES", "40"))):
        dtype = [torch.float32, torch.float16, torch.bfloat16][case % 3]
        H = int(rng.integers(2, 65))
        W = int(rng.integers(2, min(64, 4096 // H) + 1))
        B, C = int(rng.integers(1, 4)), int(rng.integers(1, 40))
        thr = int(rng.integers(1, min(H, W) // 2 + 1))
        scale = float(rng.uniform(-1.5, 2.0))
        x = (torch.randn(B, C, H, W) * float(rng.uniform(0.2, 3.0)) + float(rng.uniform(-1, 1))).to(dtype)
        ref = IO.fourier_filter_np(x.float().numpy(), thr, scale)
        y = freeu.fourier_filter(x.cuda(), thr, scale)
        err = np.abs(y.float().cpu().numpy().astype(np.float64) - ref).max()
        amax = max(1.0, np.abs(ref).max(), float(x.float().abs().max()))
        tol = {torch.float32: 4e-6, torch.float16: 2.0 ** -11 + 4e-6, torch.bfloat16: 2.0 ** -8 + 4e-6}[dtype] * amax
        assert err <= tol, (case, (B, C, H, W), thr, scale, dtype, err, tol)
