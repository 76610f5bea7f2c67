"""N > 1 path on CPU: world_size-2 gloo processes (what RCCL does over xGMI on the GPU box)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from instantrestore_amd.sharding import (gather_identities, run_sharded, run_sharded_images, scatter_identities, shard_range,
                                         shard_sizes)


def test_shard_ranges_cover_everything():
    for total in (0, 1, 5, 8, 64, 67):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert sum(shard_sizes(total, world)) == total
            assert max(shard_sizes(total, world)) - min(shard_sizes(total, world)) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        n_refs, img = 3, (3, 8, 8)
        degraded = torch.randn(total, *img) if rank == 0 else None
        refs = torch.randn(total, n_refs, *img) if rank == 0 else None
        seen = {}

        def step(d, r):  # stands in for the per-rank hot path: any per-identity function
            seen["n"] = d.shape[0]
            assert r.shape[:2] == (d.shape[0], n_refs)
            return d * 2.0 + r.sum(dim=1)

        out = run_sharded(step, degraded, refs, total, img, n_refs, torch.float32, torch.device("cpu"))
        lo, hi = shard_range(total, world, rank)
        ok = seen["n"] == hi - lo
        if rank == 0:
            want = degraded * 2.0 + refs.sum(dim=1)
            ok = ok and out is not None and torch.equal(out, want)
        else:
            ok = ok and out is None
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        # scatter/gather round trip on an uneven split
        x = torch.arange(total * 4, dtype=torch.float32).reshape(total, 4) if rank == 0 else None
        sh = scatter_identities(x, total, (4,), torch.float32, torch.device("cpu"))
        back = gather_identities(sh, total)
        if rank == 0:
            ok = ok and torch.equal(back, x)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_two_rank_scatter_step_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


# ---- the uint8 image path: preprocess output = send buffers, tensor2im before the gather --------
def _cpu_preprocess(flat):      # stand-in for the HIP Lanczos preprocessor (no CPU fallback): equal-sized images only
    x = torch.stack([im.permute(2, 0, 1).float() / 255.0 for im in flat])
    return ((x - 0.5) / 0.5).to(torch.float16)


def _cpu_tensor2im(x):          # the reference's tensor2im arithmetic (vis_utils.py:14-23) in the tensor's dtype
    v = ((x * 0.5 + 0.5).clamp(0, 1) * 255).to(torch.uint8)
    return v.permute(0, 2, 3, 1).contiguous()


def _image_worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_refs, S = 2, 8
        g = torch.Generator().manual_seed(1)
        images = [[torch.randint(0, 256, (S, S, 3), generator=g, dtype=torch.uint8) for _ in range(1 + n_refs)] for _ in range(total)]
        sent = {"n": 0, "dtypes": set()}
        real_batch = dist.batch_isend_irecv

        def counting(ops):
            sent["n"] += len(ops)
            sent["dtypes"].update(op.tensor.dtype for op in ops)
            return real_batch(ops)

        dist.batch_isend_irecv = counting
        step = lambda d, r: (d.float() * 0.5 + r.float().mean(dim=1) * 0.5).to(torch.float16)
        out = run_sharded_images(step, images if rank == 0 else None, total, n_refs, S, torch.float16, torch.device("cpu"),
                                 preprocess=_cpu_preprocess, to_image=_cpu_tensor2im)
        dist.batch_isend_irecv = real_batch
        ok = True
        if rank == 0:
            packed = _cpu_preprocess([im for ident in images for im in ident]).view(total, 1 + n_refs, 3, S, S)
            want = _cpu_tensor2im(step(packed[:, 0], packed[:, 1:]))
            ok = out is not None and out.dtype == torch.uint8 and tuple(out.shape) == (total, S, S, 3) and torch.equal(out, want)
            ok = ok and sent["n"] == 2 * (world - 1)           # ONE send per peer out, ONE receive per peer back
        else:
            ok = out is None and sent["n"] == 2
        ok = ok and sent["dtypes"] == {torch.float16, torch.uint8}   # normalised tensors out, uint8 pixels back
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_rank_uint8_image_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_image_worker, args=(r, 2, port, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


def test_cfg3_partition_eight_ranks_sixty_four_identities():
    """cfg 3 (BASELINE.json: batch = 64 identities sharded over 8 MI355X): the same scatter -> step -> gather with world size 8 and
    64 identities, eight gloo processes on this CPU - every rank gets exactly 8 contiguous identities (contiguous
    B / G split, one peer per link, no data-path collective), rank 0 gets all 64 outputs back in order.  What the 8-GPU node
    adds is RCCL over xGMI under the same calls; no such node has been available."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, total = 8, 64
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(r for r, _ in results) == list(range(world))
    assert all(ok for _, ok in results), results
    assert shard_sizes(total, world) == [8] * 8
