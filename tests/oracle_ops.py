"""TEST-ONLY stand-in for ``instantrestore_amd.ops`` backed by the oracle (CPU, float32/64).

The product has no CPU path.  To exercise the HOST logic (processors, registration, harvest,
kwargs plumbing) on the GPU-less CI box, tests monkeypatch the ``_ops`` name of
``instantrestore_amd.attn_processors`` / ``kv_harvest`` with this module.  Same function
signatures as ``instantrestore_amd.ops``."""
import numpy as np
import torch

from oracle import shared_attn_oracle as O

HEAD_DIM = 64
STATS_MAX_CHUNKS = 256
ADAIN_EPS = 1e-5
CALLS = []  # (name, info) log so tests can assert which entry points the processors used


def _np(t):
    return None if t is None else t.detach().float().cpu().numpy().astype(np.float64)


def shared_attention(q, k_self, v_self, ref_k=None, ref_v=None, *, heads, scale, include_self=True,
                     adain=None, return_lse=False, valid_refs=None, return_mass=False):
    """``valid_refs``: the caller's promise that references n >= valid_refs[b] are all-zero; the oracle walks them like any
    other (that IS the semantics: zeroed, not masked) and the stand-in only checks the promise"""
    CALLS.append(("shared_attention", dict(include_self=include_self, adain=adain is not None,
                                           n_refs=0 if ref_k is None else ref_k.shape[1],
                                           valid_refs=None if valid_refs is None else [int(x) for x in valid_refs.tolist()])))
    if valid_refs is not None:
        for b, nv in enumerate(valid_refs.tolist()):
            assert float(ref_k[b, int(nv):].abs().sum()) == 0.0 and float(ref_v[b, int(nv):].abs().sum()) == 0.0, "valid_refs promise broken"
    qn, kn, vn, rkn, rvn = map(_np, (q, k_self, v_self, ref_k, ref_v))
    if adain is not None:  # apply the affine the stats kernel would have produced
        a, b = (_np(t).reshape(rvn.shape[0], rvn.shape[1], 1, -1) for t in adain)
        rvn = rvn * a + b
    out, probs = O.shared_attention_np(qn, kn, vn, rkn, rvn, heads, scale, False, include_self, return_probs=True)
    out = torch.from_numpy(out).to(q.dtype)
    res = (out,)
    if return_lse:
        qh = O.head_to_batch_dim_np(qn, heads)
        ek, _ = O.extended_kv_np(kn, vn, rkn, rvn, heads, False, include_self)
        s = np.matmul(qh, ek.transpose(0, 2, 1)) * scale
        m = s.max(-1)
        lse = (m + np.log(np.exp(s - m[..., None]).sum(-1))).reshape(q.shape[0], heads, q.shape[1])
        res += (torch.from_numpy(lse).float(),)
    if return_mass:   # ABI v9 seg_mass: the probabilities summed per K/V segment
        res += (_segment_sums(probs, kn, rkn, include_self),)
    return res if len(res) > 1 else out


def _segment_sums(p, kn, rkn, include_self):
    edges = [0] + ([kn.shape[1]] if include_self else [])
    for n in range(0 if rkn is None else rkn.shape[1]):
        edges.append(edges[-1] + rkn.shape[2])
    mass = np.stack([p[..., a:b].sum(-1) for a, b in zip(edges[:-1], edges[1:])], axis=-1)
    return torch.from_numpy(mass).float()


def attn_probs(q, k_self, ref_k, lse, *, heads, scale, include_self=True, q_prescaled=False):
    CALLS.append(("attn_probs", {}))
    qn, kn, rkn = map(_np, (q, k_self, ref_k))
    _, p = O.shared_attention_np(qn, kn, kn, rkn, rkn, heads, scale, False, include_self, return_probs=True)
    return torch.from_numpy(p).to(q.dtype)


def attn_segment_mass(q, k_self, ref_k, lse, *, heads, scale, include_self=True, q_prescaled=False):
    CALLS.append(("attn_segment_mass", {}))
    qn, kn, rkn = map(_np, (q, k_self, ref_k))
    _, p = O.shared_attention_np(qn, kn, kn, rkn, rkn, heads, scale, False, include_self, return_probs=True)
    return _segment_sums(p, kn, rkn, include_self)


def adain_stats(v_self, ref_v, *, heads, eps=ADAIN_EPS):
    CALLS.append(("adain_stats", {}))
    a, b = O.adain_affine_np(_np(v_self), _np(ref_v), heads)
    B, N = ref_v.shape[:2]
    return (torch.from_numpy(a).float().reshape(B, N, heads, 64), torch.from_numpy(b).float().reshape(B, N, heads, 64))


def token_stats(x, *, heads):
    CALLS.append(("token_stats", {}))
    xn = _np(x)
    B, M, L, C = xn.shape
    mean = xn.mean(axis=2)
    std = np.sqrt(((xn - mean[:, :, None]) ** 2).sum(axis=2) / (L - 1))
    return (torch.from_numpy(mean).float().reshape(B, M, heads, 64), torch.from_numpy(std).float().reshape(B, M, heads, 64))


def adain_apply(x, a, b, *, heads):
    CALLS.append(("adain_apply", {}))
    B, N, L, C = x.shape
    y = _np(x) * _np(a).reshape(B, N, 1, C) + _np(b).reshape(B, N, 1, C)
    return torch.from_numpy(y).to(x.dtype)


def zero_invalid_refs(k, v, valid_indices, *, heads):
    CALLS.append(("zero_invalid_refs", {}))
    for b, idx in enumerate(torch.as_tensor(valid_indices).tolist()):
        k[b, int(idx):] = 0
        v[b, int(idx):] = 0


def linear_supported(x, weight, bias):
    """the stand-in has no GEMM of its own: the processors keep ``F.linear`` on CPU"""
    return False
