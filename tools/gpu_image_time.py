"""Timing of the image-side kernels (HBM/latency-bound byte work): Lanczos preprocess of one cfg-2
batch (8 identities x (1 degraded + 4 references) = 40 images) and the FreeU skip filter at the
UNet's two FreeU resolutions; the oracle (numpy) / Pillow-free CPU cost is bench.py's business,
here only device time and achieved bytes/s."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from instantrestore_amd import freeu
from instantrestore_amd.preprocess import LanczosPreprocessor, resize_output_size


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, (h, w) in (("1024x1024 sources", (1024, 1024)), ("3000x4000 photos", (3000, 4000)), ("512x512 (already sized)", (512, 512))):
    imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(40)]
    pre = LanczosPreprocessor(512, torch.float16)
    ms_host = timeit(lambda: pre(imgs))
    descs, dev = pre.prepare(imgs)
    ms = timeit(lambda: pre.run(descs, dev))
    oh, ow = resize_output_size(h, w, 512)
    # algorithmic bytes: the source region under the crop once + the fp16 output
    src = 40 * (h * min(w, int(512 * w / ow) + 1)) * 3 if ow >= oh else 40 * (min(h, int(512 * h / oh) + 1) * w) * 3
    byt = src + 40 * 3 * 512 * 512 * 2
    print(f"preprocess 40 x {name}: {ms*1e3:8.1f} us  {byt/ms/1e6:8.1f} GB/s algorithmic ({byt/1e6:.1f} MB); with host descriptor build {ms_host*1e3:.1f} us")

try:   # context only: the CPU path the reference runs per image (Pillow resize + numpy normalise)
    import time
    from PIL import Image
    a = np.random.default_rng(0).integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    im = Image.fromarray(a, "RGB")
    t0 = time.perf_counter()
    for _ in range(5):
        r = np.asarray(im.resize((512, 512), Image.LANCZOS), dtype=np.float32) / 255.0
        r = (r - 0.5) / 0.5
    print(f"host Pillow {Image.__version__ if hasattr(Image, '__version__') else ''} 1024x1024 -> 512: {(time.perf_counter()-t0)/5*1e3:.2f} ms per image (1 thread)")
except ImportError:
    pass

for shape in ((8, 1280, 8, 8), (8, 1280, 16, 16), (32, 1280, 16, 16), (16, 1280, 32, 32)):
    x = torch.randn(shape, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: freeu.fourier_filter(x, 1, 0.9))
    byt = 2 * x.numel() * 2
    def ref():
        xf = x.float()
        f = torch.fft.fftshift(torch.fft.fftn(xf, dim=(-2, -1)), dim=(-2, -1))
        mask = torch.ones_like(xf)
        mask[..., shape[2] // 2 - 1:shape[2] // 2 + 1, shape[3] // 2 - 1:shape[3] // 2 + 1] = 0.9
        return torch.fft.ifftn(torch.fft.ifftshift(f * mask, dim=(-2, -1)), dim=(-2, -1)).real.to(x.dtype)
    ms_ref = timeit(ref)
    err = (ref().float() - freeu.fourier_filter(x, 1, 0.9).float()).abs().max().item()
    print(f"freeu filter {shape}: {ms*1e3:7.1f} us  {byt/ms/1e6:7.1f} GB/s | torch.fft sequence {ms_ref*1e3:7.1f} us  (max diff {err:.1e})")
