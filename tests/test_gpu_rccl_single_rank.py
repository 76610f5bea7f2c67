"""RCCL on the one GPU of this box: a REAL one-rank ``nccl`` process group - the communicator the N > 1
job builds per rank - carries the bench's control plane (barrier, MAX-over-ranks reduction, bench.py) and the scatter /
gather leg of SURVEY.md section 8e (``sharding.run_sharded_images`` with both transfers as ``ncclSend`` + ``ncclRecv`` to
this rank inside one group), and the pixels that come back are the bytes of the slicing path.  No xGMI link is involved and
no scaling is measured; what this buys is that the first 8-GPU run is not also the first time RCCL executes under this code.
Runs in a subprocess: a process group is process-wide state the rest of the suite must not inherit."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import json, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
from instantrestore_amd import sharding
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
rep = bench.init_single_rank_nccl(dev)
assert rep["ok"], rep
res = {"init": rep, "backend": dist.get_backend(), "world": dist.get_world_size()}
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
res["max"] = float(t.item())
g = torch.Generator().manual_seed(11)
total, N, px = 3, 2, 64
images = [[torch.randint(0, 256, (90 + 7 * i, 80 + 5 * j, 3), generator=g, dtype=torch.uint8).to(dev) for j in range(1 + N)] for i in range(total)]
step = lambda d, r: d * 0.5 + r[:, 0] * 0.25
a = sharding.run_sharded_images(step, images, total, N, px, torch.float16, dev, loopback=True)
b = sharding.run_sharded_images(step, images, total, N, px, torch.float16, dev)
torch.cuda.synchronize()
res["shape"] = list(a.shape)
res["dtype"] = str(a.dtype)
res["equal"] = bool(torch.equal(a, b))
x = torch.randn(5, 7, device=dev)
res["loopback_tensor"] = bool(torch.equal(sharding.scatter_identities(x, 5, (7,), x.dtype, dev, loopback=True), x))
dist.destroy_process_group()
print(json.dumps(res))
'''


def test_one_rank_rccl_communicator_carries_control_plane_and_scatter_gather():
    assert torch.cuda.is_available()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT, REPO], capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    # (librccl prints "Librccl path : ..." through C stdio, flushed at exit: the JSON line is not necessarily the last one)
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["backend"] == "nccl" and res["world"] == 1 and res["init"]["ok"]
    assert res["max"] == 3.5
    assert res["shape"] == [3, 64, 64, 3] and res["dtype"] == "torch.uint8"
    assert res["equal"] and res["loopback_tensor"]
