"""rocprofv3 target: the Lanczos preprocess of 40 images of one size (argv: H W), nothing else."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from instantrestore_amd.preprocess import LanczosPreprocessor
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1024)
imgs = [torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, device="cuda") for _ in range(40)]
pre = LanczosPreprocessor(512, torch.float16)
for _ in range(10):
    pre(imgs)
torch.cuda.synchronize()
