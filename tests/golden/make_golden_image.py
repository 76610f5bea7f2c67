#!/usr/bin/env python3
"""Golden vectors for the caller's input transform (test.py:54-59), made in the build container
by the third-party code the reference calls: ``PIL.Image.resize(..., LANCZOS)`` (Pillow is
installed here; torchvision is not, its Resize/CenterCrop size rules are restated in
oracle/image_oracle.py).  Data only: input bytes and Pillow's output bytes.

    python tests/golden/make_golden_image.py      # writes tests/golden/image_golden.npz
"""
import json
import os

import numpy as np
import PIL
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))

# (in_h, in_w, Resize size): out size follows torchvision's rule, computed by the oracle's restatement
CASES = [
    (37, 53, 32), (96, 64, 32), (130, 200, 64), (64, 64, 64), (50, 70, 96), (33, 91, 32),
    (257, 255, 64), (40, 41, 40), (64, 300, 48), (300, 64, 48), (171, 128, 64), (300, 420, 128),
]


def main():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle.image_oracle import resize_output_size
    rng = np.random.default_rng(20240917)
    arrays, manifest = {}, []
    for i, (h, w, size) in enumerate(CASES):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if i % 2 == 0:  # smooth content with saturated regions: exercises the clip8 at both ends
            yy, xx = np.mgrid[0:h, 0:w]
            img[..., 0] = ((np.sin(yy / 5.0) + np.cos(xx / 3.0)) * 120 + 128).clip(0, 255).astype(np.uint8)
            img[..., 1] = np.where((yy // 8 + xx // 8) % 2 == 0, 255, 0).astype(np.uint8)
        out_h, out_w = resize_output_size(h, w, size)
        ref = np.asarray(Image.fromarray(img, "RGB").resize((out_w, out_h), Image.LANCZOS))
        arrays[f"c{i}_in"] = img
        arrays[f"c{i}_resized"] = ref
        manifest.append({"id": f"c{i}", "in_h": h, "in_w": w, "size": size, "out_h": out_h, "out_w": out_w})
    arrays["manifest"] = np.frombuffer(json.dumps({"pillow": PIL.__version__, "cases": manifest}).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "image_golden.npz"), **arrays)
    print("wrote", len(manifest), "cases, Pillow", PIL.__version__)


if __name__ == "__main__":
    main()
