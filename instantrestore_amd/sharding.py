"""Multi-GPU execution of the hot path: independent identities, one process per GPU.

Every identity ``b`` is independent through the whole pipeline - the shared attention never
mixes batch entries (``ref_keys[b]`` only feeds sample ``b``, attn_processors.py:238-241) - so
the path shards embarrassingly: a contiguous split of the identities over the ranks, weights
replicated, NO collective on the data path.  The reference has no
inference-time distribution at all (its only multi-GPU code is training DDP through accelerate,
coach.py:52-61); this is the MI355X deployment shape for it.

The only traffic is the optional batch scatter / output gather when one rank owns the inputs.
Over xGMI (point-to-point, 7 links per GPU) that is issued as one grouped batch of send/recv
pairs, so every link carries exactly one peer's shard - never a ring.  With ``backend="nccl"``
these are RCCL ``ncclSend/ncclRecv`` inside one group; with ``gloo`` (CPU tests) plain TCP.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """contiguous split; the first ``total % world`` ranks get one extra identity"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int) -> List[int]:
    return [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]


def _world(group) -> Tuple[int, int]:
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _loopback(t: torch.Tensor, group) -> torch.Tensor:
    """one rank, ``loopback``: the shard travels through ONE grouped send/recv pair addressed to this rank itself - with the
    ``nccl`` backend an ``ncclSend`` + ``ncclRecv`` inside one RCCL group on the real communicator.  No reference counterpart
    and no use in production (a one-rank job slices); it exists so that the transfer code path of the N > 1 job - communicator,
    group launch, stream ordering - has executed on every box the single-GPU bench and tests run on."""
    if dist.get_backend(group) != "nccl":
        raise RuntimeError("loopback transfers need the nccl (RCCL) backend: gloo has no pair to the calling rank itself")
    src = t.contiguous()
    dst = torch.empty_like(src)
    me = dist.get_rank(group)
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, me, group), dist.P2POp(dist.irecv, dst, me, group)]):
        req.wait()
    return dst


def scatter_identities(full: Optional[torch.Tensor], total: int, tail_shape: Sequence[int], dtype: torch.dtype,
                       device: torch.device, src: int = 0, group=None, loopback: bool = False) -> torch.Tensor:
    """Rank ``src`` holds ``full`` of shape ``(total, *tail_shape)``; every rank returns its
    contiguous shard.  One grouped batch of point-to-point transfers (one peer per link)."""
    world, rank = _world(group)
    lo, hi = shard_range(total, world, rank)
    if world == 1:
        if loopback and dist.is_available() and dist.is_initialized():
            return _loopback(full[lo:hi], group)
        return full[lo:hi]
    ops, keep = [], []
    if rank == src:
        if full is None or full.shape[0] != total:
            raise ValueError("source rank must pass the full batch")
        for r in range(world):
            if r == src:
                continue
            rlo, rhi = shard_range(total, world, r)
            if rhi > rlo:
                piece = full[rlo:rhi].contiguous()   # a leading-axis slice of a contiguous batch IS contiguous: no copy,
                keep.append(piece)                   # the producer's output buffer is the send buffer
                ops.append(dist.P2POp(dist.isend, piece, r, group))
        mine = full[lo:hi].clone()
    else:
        mine = torch.empty((hi - lo, *tail_shape), dtype=dtype, device=device)
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, mine, src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return mine


def gather_identities(shard: torch.Tensor, total: int, dst: int = 0, group=None, loopback: bool = False) -> Optional[torch.Tensor]:
    """Inverse of :func:`scatter_identities`: rank ``dst`` returns ``(total, ...)``, others None."""
    world, rank = _world(group)
    if world == 1:
        if loopback and dist.is_available() and dist.is_initialized():
            return _loopback(shard, group)
        return shard
    ops = []
    out = None
    if rank == dst:
        out = torch.empty((total, *shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        for r in range(world):
            rlo, rhi = shard_range(total, world, r)
            if r == dst:
                out[rlo:rhi].copy_(shard)
            elif rhi > rlo:
                ops.append(dist.P2POp(dist.irecv, out[rlo:rhi], r, group))
    elif shard.shape[0] > 0:
        ops.append(dist.P2POp(dist.isend, shard.contiguous(), dst, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return out


def run_sharded(step_fn, degraded: Optional[torch.Tensor], refs: Optional[torch.Tensor], total: int,
                img_shape: Sequence[int], n_refs: int, dtype: torch.dtype, device: torch.device, group=None):
    """scatter -> ``step_fn(degraded_shard, refs_shard)`` on every rank -> gather on rank 0.

    ``degraded`` ``(total, *img_shape)`` and ``refs`` ``(total, n_refs, *img_shape)`` live on
    rank 0 (others pass ``None``); the batch layout is the caller's (test.py:79-111)."""
    d = scatter_identities(degraded, total, tuple(img_shape), dtype, device, 0, group)
    r = scatter_identities(refs, total, (n_refs, *img_shape), dtype, device, 0, group)
    out = step_fn(d, r)
    return gather_identities(out, total, 0, group)


def run_sharded_images(step_fn, images, total: int, n_refs: int, size: int, dtype: torch.dtype, device: torch.device,
                       preprocess=None, to_image=None, group=None, loopback: bool = False):
    """The caller's whole per-batch data path with the image kernels fused into the shard transfers:

    rank 0 holds ``images``: ``total`` identities, each a sequence of ``1 + n_refs`` ``uint8`` ``(H, W, 3)`` tensors
    (degraded image first, then its references; ragged sizes allowed).  ``preprocess`` (default
    :class:`instantrestore_amd.preprocess.LanczosPreprocessor` - Pillow-exact resize, crop, normalise, cast on the
    device) is launched ONCE over the identity-major flat list and writes ``(total, 1 + n_refs, 3, size, size)``: an
    identity's degraded image and references are adjacent, so every peer's shard is ONE contiguous slice of that output -
    the kernel's output buffer is the send buffer, one transfer per peer (the two-tensor form sends two).  Every rank runs
    ``step_fn(degraded (b, 3, S, S), refs (b, N, 3, S, S)) -> (b, 3, S, S)`` on its shard and ``to_image`` (default
    ``ops.tensor2im_u8``: the reference's ``tensor2im``, vis_utils.py:14-23, on the device) BEFORE the gather, so ``uint8``
    ``(b, S, S, 3)`` pixels travel back - a third of the fp16 tensor's bytes.  Rank 0 returns ``(total, S, S, 3)`` uint8,
    the others ``None``.  ``preprocess`` / ``to_image`` are injectable for the CPU (gloo) tests: the HIP ones have no CPU
    fallback.  ``loopback``: on ONE rank with an initialised process group, send both transfers through the communicator
    to this rank itself (:func:`_loopback`) instead of slicing - same bytes out."""
    world, rank = _world(group)
    if preprocess is None:
        from .preprocess import LanczosPreprocessor
        preprocess = LanczosPreprocessor(size, dtype)
    if to_image is None:
        from . import ops as _ops
        to_image = _ops.tensor2im_u8
    packed = None
    if rank == 0:
        if images is None or len(images) != total or any(len(ident) != 1 + n_refs for ident in images):
            raise ValueError("rank 0 must pass `total` identities of 1 + n_refs images each")
        flat = [im for ident in images for im in ident]                  # identity-major: shards are contiguous
        packed = preprocess(flat).view(total, 1 + n_refs, 3, size, size)
    shard = scatter_identities(packed, total, (1 + n_refs, 3, size, size), dtype, device, 0, group, loopback)
    out = step_fn(shard[:, 0], shard[:, 1:])
    pixels = to_image(out)                                               # (b, S, S, 3) uint8
    if pixels.dtype != torch.uint8:
        raise TypeError("to_image must return uint8 pixels")
    return gather_identities(pixels, total, 0, group, loopback)
