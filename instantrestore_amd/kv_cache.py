"""Per-identity cache of harvested reference K/V.

The reference recomputes the K/V of the reference faces for every restored frame
(``Pix2Pix_Turbo.forward`` -> ``get_conditioning_keys_values``, pix2pix_turbo.py:297-298): N of
the N+1 UNet forwards, N of the N+1 VAE encodes and an unused VAE decode (:277-279) per frame,
although the references of one identity never change.  This cache keeps, per identity, the nine
``(1, N, L, C)`` key tensors and nine value tensors produced by
:func:`instantrestore_amd.kv_harvest.get_conditioning_keys_values`; a batch is assembled by
concatenation along the identity axis (one copy of the cached tensors per batch, instead of N UNet
forwards per identity).  Behaviour-preserving for inference callers: the K/V handed to the main
UNet are the same tensors the reference would have recomputed (up to the reference's own RNG:
``randn_like`` noise at t=1, pix2pix_turbo.py:248 - caching freezes one draw).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Callable, Hashable, List, Sequence, Tuple

import torch

KV = Tuple[List[torch.Tensor], ...]   # (keys, values) or (keys, values, stats): stats[l] = (mean, std) fp32 (1, N, H, 64)


class ReferenceKVCache:
    def __init__(self, max_identities: int = 64):
        if max_identities < 1:
            raise ValueError("max_identities must be >= 1")
        self.max_identities = max_identities
        self._store: "OrderedDict[Hashable, KV]" = OrderedDict()
        self.hits = 0
        self.misses = 0

    def __len__(self) -> int:
        return len(self._store)

    def __contains__(self, identity: Hashable) -> bool:
        return identity in self._store

    def get_or_compute(self, identity: Hashable, compute: Callable[[], KV]) -> KV:
        """``compute()`` must return ``(keys, values)`` for ONE identity: lists of ``(1, N, L, C)`` - or ``(keys, values,
        stats)`` as ``get_conditioning_keys_values(..., with_stats=True)`` does: the AdaIN content statistics of the reference
        V's (``(mean, std)`` per layer, fp32 ``(1, N, H, 64)``) are constant per identity too and are cached with them."""
        if identity in self._store:
            self._store.move_to_end(identity)
            self.hits += 1
            return self._store[identity]
        self.misses += 1
        res = compute()
        if not isinstance(res, (tuple, list)) or len(res) not in (2, 3):
            # harvest_reference_kv(with_events=True) returns (keys, values, events[, stats]): events are not cacheable and
            # must not be mistaken for statistics - accept the two documented forms only
            raise ValueError("compute() must return (keys, values) or (keys, values, stats); harvest with with_events=False")
        keys, values = res[0], res[1]
        stats = res[2] if len(res) > 2 else None
        if stats is not None:      # statistics still in the GEMM-partials form (ops.RefStatsPartials): a cache entry holds them finished
            stats = [st.finished() if hasattr(st, "finished") else st for st in stats]
        if len(keys) != len(values) or any(k.shape[0] != 1 or k.shape != v.shape for k, v in zip(keys, values)):
            raise ValueError("compute() must return matching lists of (1, N, L, C) tensors")
        if stats is not None:
            def _is_stat(st):
                return st is None or (isinstance(st, (tuple, list)) and len(st) == 2 and
                                      all(isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.dim() == 4 and t.shape[0] == 1 for t in st))
            if len(stats) != len(keys) or not all(_is_stat(st) for st in stats):
                raise ValueError("stats must be a per-layer list of None or (mean, std) fp32 tensors of shape (1, N, H, 64)")
        # compact copies: the harvested tensors are strided views of the capture layers' fused (B*N, L, 3C) projection
        # output - caching the views would pin that whole buffer (dead Q third, every other identity of the batch) per
        # entry and eviction would free nothing.  One entry = 2 * 9 layers * N * L * C * 2 bytes.
        entry = ([k.detach().clone(memory_format=torch.contiguous_format) for k in keys],
                 [v.detach().clone(memory_format=torch.contiguous_format) for v in values])
        if stats is not None:
            entry = entry + ([None if st is None else (st[0].detach().clone(), st[1].detach().clone()) for st in stats],)
        self._store[identity] = entry
        while len(self._store) > self.max_identities:
            self._store.popitem(last=False)
        return entry

    def assemble(self, identities: Sequence[Hashable]) -> KV:
        """``(B, N, L, C)`` lists for a batch of cached identities (raises KeyError on a miss)."""
        entries = [self._store[i] for i in identities]
        for i in identities:
            self._store.move_to_end(i)
        n_layers = len(entries[0][0])
        keys = [torch.cat([e[0][l] for e in entries], dim=0) for l in range(n_layers)]
        values = [torch.cat([e[1][l] for e in entries], dim=0) for l in range(n_layers)]
        with_stats = [len(e) > 2 for e in entries]
        if any(with_stats) and not all(with_stats):
            raise ValueError("assemble(): some of these identities were cached with AdaIN content statistics and some without; "
                             "cache them the same way (the statistics would otherwise be dropped silently)")
        if all(with_stats):
            stats = [None if any(e[2][l] is None for e in entries) else
                     (torch.cat([e[2][l][0] for e in entries], dim=0), torch.cat([e[2][l][1] for e in entries], dim=0))
                     for l in range(n_layers)]
            return keys, values, stats
        return keys, values

    def nbytes(self, identity: Hashable) -> int:
        """device bytes held for one cached identity"""
        e = self._store[identity]
        extra = [t for st in (e[2] if len(e) > 2 else []) if st is not None for t in st]
        return sum(t.untyped_storage().nbytes() for t in e[0] + e[1] + extra)

    def invalidate(self, identity: Hashable = None) -> None:
        if identity is None:
            self._store.clear()
        else:
            self._store.pop(identity, None)
