import pytest
import torch

from instantrestore_amd.kv_cache import ReferenceKVCache


def _entry(seed, n_layers=3, N=2):
    g = torch.Generator().manual_seed(seed)
    ks = [torch.randn(1, N, 4 * (l + 1), 64, generator=g) for l in range(n_layers)]
    vs = [torch.randn(1, N, 4 * (l + 1), 64, generator=g) for l in range(n_layers)]
    return ks, vs


def test_cache_hits_misses_lru_and_assembly():
    cache = ReferenceKVCache(max_identities=2)
    calls = []

    def compute(seed):
        def f():
            calls.append(seed)
            return _entry(seed)
        return f

    a = cache.get_or_compute("alice", compute(1))
    assert cache.get_or_compute("alice", compute(99)) is a and calls == [1]
    cache.get_or_compute("bob", compute(2))
    keys, vals = cache.assemble(["bob", "alice"])
    assert [tuple(k.shape) for k in keys] == [(2, 2, 4, 64), (2, 2, 8, 64), (2, 2, 12, 64)]
    assert torch.equal(keys[0][1:], a[0][0]) and torch.equal(vals[2][0:1], cache.get_or_compute("bob", compute(0))[1][2])
    cache.get_or_compute("carol", compute(3))          # evicts the least recently used: alice
    assert "alice" not in cache and "bob" in cache and len(cache) == 2
    assert (cache.hits, cache.misses) == (2, 3)
    with pytest.raises(KeyError):
        cache.assemble(["alice"])
    with pytest.raises(ValueError):
        cache.get_or_compute("bad", lambda: ([torch.zeros(2, 1, 4, 64)], [torch.zeros(2, 1, 4, 64)]))
    cache.invalidate("bob")
    assert "bob" not in cache
    cache.invalidate()
    assert len(cache) == 0
