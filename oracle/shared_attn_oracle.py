"""ORACLE - test infrastructure only, never the product path.

CPU restatement of the one hot path this repository accelerates: InstantRestore's shared-image
(extended) self-attention plus the AdaIN value injection, as computed by
``/root/reference/face_replace/models/attn_processors.py`` (``SharedAttnProcessor.forward``
``:193-279``, ``adain`` ``:7-18``, ``AttnProcessor.forward`` ``:34-97``) on top of the
``diffusers==0.24.0`` ``Attention`` helpers it calls (third-party, pinned at
``environment_new.yml:89``, source not under the reference tree; its published algorithm -
``head_to_batch_dim`` = reshape/permute, ``get_attention_scores`` = ``baddbmm(alpha=scale)``
-> ``softmax(-1)``, SURVEY.md Appendix A - is restated here).

Pinning: the reference has no tests and no golden vectors of its own.
This oracle is pinned against outputs of the reference itself, produced in the build
container by importing the reference's ``attn_processors.py`` (``tests/golden/make_golden.py``
-> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).  The diffusers
``Attention`` seam is the one assumption that cannot be pinned offline ("parity unpinned" at
that seam only).

Who may import this file: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` - as the checker / the timed CPU baseline, never as a fallback of the product.

Two implementations live here:

* ``*_np``   : numpy, float64 by default - the truth used for parity (small/medium sizes);
* ``*_port`` : torch-CPU float32, same operator sequence as the reference (materialised
               probabilities, per-reference head-split copies, ``cat``) - this is what
               ``bench.py`` times as ``cpu_baseline.kind == "port"`` and what full-size GPU
               parity is sampled against.
"""
from __future__ import annotations

import numpy as np

ADAIN_EPS = 1e-5  # attn_processors.py:10 and :245 (added to BOTH standard deviations)


# --------------------------------------------------------------------------------------------
# numpy truth
# --------------------------------------------------------------------------------------------
def head_to_batch_dim_np(t: np.ndarray, heads: int) -> np.ndarray:
    """(B, S, H*D) -> (B*H, S, D); diffusers Attention.head_to_batch_dim."""
    b, s, c = t.shape
    return t.reshape(b, s, heads, c // heads).transpose(0, 2, 1, 3).reshape(b * heads, s, c // heads)


def batch_to_head_dim_np(t: np.ndarray, heads: int) -> np.ndarray:
    """(B*H, S, D) -> (B, S, H*D); diffusers Attention.batch_to_head_dim."""
    bh, s, d = t.shape
    return t.reshape(bh // heads, heads, s, d).transpose(0, 2, 1, 3).reshape(bh // heads, s, d * heads)


def token_stats_np(x: np.ndarray):
    """mean and UNBIASED std over the token axis (dim=1), keepdim; torch defaults used at
    attn_processors.py:9-10 and :244-245.  A single token gives NaN std, as torch does."""
    mean = x.mean(axis=1, keepdims=True)
    n = x.shape[1]
    with np.errstate(invalid="ignore", divide="ignore"):
        var = ((x - mean) ** 2).sum(axis=1, keepdims=True) / np.float64(n - 1)
    return mean, np.sqrt(var).astype(x.dtype)


def adain_np(content: np.ndarray, style_mean: np.ndarray, style_std: np.ndarray) -> np.ndarray:
    """attn_processors.py:7-18.  ``style_std`` already carries its +1e-5 (call site :245)."""
    c_mean, c_std = token_stats_np(content)
    c_std = c_std + ADAIN_EPS
    return (content - c_mean) / c_std * style_std + style_mean


def adain_affine_np(v_self: np.ndarray, ref_v: np.ndarray, heads: int):
    """The same AdaIN written as a per-(b, n, channel) affine ``x*a + b`` (what the HIP stats
    kernel emits): a = (sigma_v+eps)/(sigma_x+eps), b = mu_v - mu_x*a.
    v_self (B, L, C), ref_v (B, N, Lr, C) -> a, b of shape (B, N, C)."""
    mu_v, sd_v = token_stats_np(v_self)  # (B,1,C)
    B, N = ref_v.shape[:2]
    a = np.empty((B, N, ref_v.shape[-1]), dtype=v_self.dtype)
    b = np.empty_like(a)
    for n in range(N):
        mu_x, sd_x = token_stats_np(ref_v[:, n])
        with np.errstate(invalid="ignore", divide="ignore"):
            a_n = (sd_v + ADAIN_EPS) / (sd_x + ADAIN_EPS)
        # content std exactly 0 (a zero-filled reference, pix2pix_turbo.py:269-273, or a constant channel): every token equals
        # the mean, adain_np() above returns exactly style_mean for ANY ratio - the affine that says so without a 1e5 * x
        # against -1e5 * mean cancellation is (0, mu_v).  adain_np(x) == x * a + b holds either way (tests/test_oracle_golden.py).
        a_n = np.where(sd_x == 0, 0.0, a_n)
        a[:, n] = a_n[:, 0]
        b[:, n] = (mu_v - mu_x * a_n)[:, 0]
    return a, b


def softmax_np(s: np.ndarray) -> np.ndarray:
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m)
    return e / e.sum(axis=-1, keepdims=True)


def extended_kv_np(k_self, v_self, ref_k, ref_v, heads, use_adain, train_input):
    """K/V of the extended sequence, head-split: attn_processors.py:232-255.
    Column order: [self tokens (only if train_input)] ++ ref 0 ++ ... ++ ref N-1."""
    key = head_to_batch_dim_np(k_self, heads)
    value = head_to_batch_dim_np(v_self, heads)
    if ref_k is None or ref_v is None:
        return key, value
    n_refs = ref_k.shape[1]  # N comes from the tensor (:240-241)
    ks = [head_to_batch_dim_np(ref_k[:, n], heads) for n in range(n_refs)]
    vs = [head_to_batch_dim_np(ref_v[:, n], heads) for n in range(n_refs)]
    if use_adain:
        s_mean, s_std = token_stats_np(value)
        s_std = s_std + ADAIN_EPS
        vs = [adain_np(v, s_mean, s_std) for v in vs]
    if train_input:
        ks, vs = [key] + ks, [value] + vs
    return np.concatenate(ks, axis=1), np.concatenate(vs, axis=1)


def shared_attention_np(q, k_self, v_self, ref_k, ref_v, heads, scale,
                        use_adain=False, train_input=True, dtype=np.float64, return_probs=False):
    """Core of SharedAttnProcessor.forward between the q/k/v projections and ``to_out``
    (attn_processors.py:232-264).  q,k_self,v_self: (B, L, C); ref_k, ref_v: (B, N, Lr, C) or
    None (plain attention, the ``self_attn_idx is None`` branch :253-255 and the whole of
    AttnProcessor :76-82).  Returns (B, L, C) [and probs (B, H, L, Lkv)]."""
    cast = lambda t: None if t is None else np.asarray(t, dtype=dtype)
    q, k_self, v_self, ref_k, ref_v = map(cast, (q, k_self, v_self, ref_k, ref_v))
    qh = head_to_batch_dim_np(q, heads)
    ek, ev = extended_kv_np(k_self, v_self, ref_k, ref_v, heads, use_adain, train_input)
    scores = np.matmul(qh, ek.transpose(0, 2, 1)) * dtype(scale)
    probs = softmax_np(scores)
    out = batch_to_head_dim_np(np.matmul(probs, ev), heads)
    if return_probs:
        B = q.shape[0]
        return out, probs.reshape(B, heads, qh.shape[1], ek.shape[1])
    return out


def shared_attn_processor_np(hidden, wq, wk, wv, wo, bo, ref_k, ref_v, heads,
                             use_adain=False, train_input=True, encoder_hidden=None,
                             dtype=np.float64, return_probs=False):
    """Whole processor (attn_processors.py:222-269): projections, core, out projection.
    Weights in torch ``nn.Linear`` layout (out_features, in_features)."""
    c = lambda t: np.asarray(t, dtype=dtype)
    hidden = c(hidden)
    enc = hidden if encoder_hidden is None else c(encoder_hidden)
    q = hidden @ c(wq).T
    k = enc @ c(wk).T
    v = enc @ c(wv).T
    d_head = q.shape[-1] // heads
    res = shared_attention_np(q, k, v, ref_k, ref_v, heads, d_head ** -0.5,
                              use_adain, train_input, dtype, return_probs)
    core, probs = res if return_probs else (res, None)
    out = core @ c(wo).T + c(bo)
    return (out, probs, (q, k, v)) if return_probs else out


def faceid_processor_np(hidden, wq, wo, bo, wp, bp, wk, wv, heads, encoder_hidden=None, dtype=np.float64):
    """``FaceIDAttnProcessor.forward`` (attn_processors.py:115-180): queries from ``attn.to_q``; keys and values from the
    processor's own ``face_projection`` (with bias) followed by ``to_k_face_embed`` / ``to_v_face_embed`` (no bias) applied to
    the encoder states - or to the hidden states themselves when there are none (:148-153); plain softmax attention
    (:159-161), ``to_out[0]`` (:164).  No reference K/V, no AdaIN: ``ref_keys`` / ``ref_values`` are ignored (:123-124)."""
    c = lambda t: np.asarray(t, dtype=dtype)
    hidden = c(hidden)
    src = (hidden if encoder_hidden is None else c(encoder_hidden)) @ c(wp).T + c(bp)
    q, k, v = hidden @ c(wq).T, src @ c(wk).T, src @ c(wv).T
    core = shared_attention_np(q, k, v, None, None, heads, (q.shape[-1] // heads) ** -0.5, False, True, dtype)
    return core @ c(wo).T + c(bo)


def zero_fill_invalid_np(ref: np.ndarray, valid_indices) -> np.ndarray:
    """pix2pix_turbo.py:269-273: refs >= valid_indices[b] are ZEROED (not masked)."""
    out = ref.copy()
    for b, idx in enumerate(valid_indices):
        out[b, int(idx):] = 0
    return out


# --------------------------------------------------------------------------------------------
# torch-CPU float32 port: same operator sequence as the reference (timed as the CPU baseline)
# --------------------------------------------------------------------------------------------
def _h2b(t, heads):
    b, s, c = t.shape
    return t.reshape(b, s, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, s, c // heads)


def _b2h(t, heads):
    bh, s, d = t.shape
    return t.reshape(bh // heads, heads, s, d).permute(0, 2, 1, 3).reshape(bh // heads, s, d * heads)


def _adain_port(content, style_mean, style_std):
    c_mean = content.mean(dim=1, keepdim=True)
    c_std = content.std(dim=1, keepdim=True) + ADAIN_EPS
    return (content - c_mean) / c_std * style_std + style_mean


def shared_attention_port(q, k_self, v_self, ref_k, ref_v, heads, scale,
                          use_adain=False, train_input=True, return_probs=False):
    """torch version of ``shared_attention_np`` that materialises the probability matrix with
    baddbmm -> softmax -> bmm exactly like the reference + diffusers do."""
    import torch

    query, key, value = _h2b(q, heads), _h2b(k_self, heads), _h2b(v_self, heads)
    if ref_k is not None and ref_v is not None:
        n_refs = ref_k.shape[1]
        ks = [_h2b(ref_k[:, n], heads) for n in range(n_refs)]
        vs = [_h2b(ref_v[:, n], heads) for n in range(n_refs)]
        if use_adain:
            s_mean = value.mean(dim=1, keepdim=True)
            s_std = value.std(dim=1, keepdim=True) + ADAIN_EPS
            vs = [_adain_port(v, s_mean, s_std) for v in vs]
        if train_input:
            ks, vs = [key] + ks, [value] + vs
        key, value = torch.cat(ks, dim=1), torch.cat(vs, dim=1)
    base = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype)
    probs = torch.baddbmm(base, query, key.transpose(-1, -2), beta=0, alpha=scale).softmax(dim=-1)
    out = _b2h(torch.bmm(probs, value), heads)
    if return_probs:
        return out, probs.reshape(q.shape[0], heads, query.shape[1], key.shape[1])
    return out


def shared_attn_processor_port(hidden, wq, wk, wv, wo, bo, ref_k, ref_v, heads,
                               use_adain=False, train_input=True):
    """Whole processor on torch-CPU fp32 (projections + core + out projection)."""
    import torch.nn.functional as F

    q, k, v = F.linear(hidden, wq), F.linear(hidden, wk), F.linear(hidden, wv)
    core = shared_attention_port(q, k, v, ref_k, ref_v, heads, (q.shape[-1] // heads) ** -0.5,
                                 use_adain, train_input)
    return F.linear(core, wo, bo)


def tensor2im_np(var: np.ndarray) -> np.ndarray:
    """face_replace/training/utils/vis_utils.py:14-23 with unnorm=True, restated on a numpy array
    of the tensor's own dtype (float16 / float32; bf16 callers pass float32 values and round with
    ``round_fn``): (3,H,W) -> uint8 (H,W,3).  Every step rounds to the array dtype like torch does."""
    v = var.copy()
    v *= np.asarray(0.5, dtype=v.dtype)
    v += np.asarray(0.5, dtype=v.dtype)
    v = np.transpose(v, (1, 2, 0)).copy()
    v[v < 0] = 0
    v[v > 1] = 1
    v *= np.asarray(255, dtype=v.dtype)
    return v.astype("uint8")
