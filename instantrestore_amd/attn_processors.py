"""Attention processors of InstantRestore, MI355X-native.

Drop-in for ``face_replace/models/attn_processors.py`` of the reference: the same six public
names, constructor signatures, ``forward`` signatures, attribute protocol
(``self_attn_idx``, ``save_self_attentions``, ``attention_probs``, ``keys``/``values``/
``reset()``) and registration functions.  The projections (``to_q/k/v``,
``to_out``) stay ``nn.Linear`` calls on the host ``attn`` object exactly as in the reference -
so peft/LoRA wrappers and autocast keep working - while everything between them, which the
reference spells as head-split copies + ``adain`` + ``cat`` + ``baddbmm``/``softmax``/``bmm``
(attn_processors.py:232-264), is ONE fused HIP kernel plus a one-pass statistics kernel
(``instantrestore_amd.ops`` -> ``include/instantrestore_hip.h``).

``attn.upcast_attention`` / ``attn.upcast_softmax`` (diffusers ``Attention``; the reference honours them through
``get_attention_scores``, attn_processors.py:257): the fused kernel ALWAYS forms the scores from the 16-bit q / k with
fp32 accumulation (products of 16-bit values are exact in fp32: the same numbers as an fp32 ``baddbmm`` of the upcast
operands) and keeps the softmax in fp32 - both flags are satisfied by construction, set or not; with them unset the
reference's own path rounds the scores to 16 bit before its softmax, which this path never does.
``tests/test_golden_r4.py`` holds both settings to the reference's outputs.

The processors own no parameters and no buffers (the reference's checkpoints are loaded with
``strict=True``, test.py:47-50).  They never fall back to torch math: CPU tensors, fp32
activations outside autocast, attention masks and missing libraries raise.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import lora_fold as _lora
from . import ops as _ops  # tests swap this module-level name for an oracle-backed stand-in


# ------------------------------------------------------------------------------------------
# adain(): attn_processors.py:7-18
# ------------------------------------------------------------------------------------------
def adain(content_features: torch.Tensor, style_mean: torch.Tensor, style_std: torch.Tensor) -> torch.Tensor:
    """Renormalise ``content_features`` (BH, L, 64) to the given style statistics.

    Same contract as the reference function: statistics over the token axis (dim=1), unbiased
    std, ``1e-5`` added to the content std here (the caller has already added it to
    ``style_std``, attn_processors.py:245).  The reduction and the application are HIP kernels;
    only the (BH, 1, 64)-sized algebra that turns four statistics into an affine is torch.
    """
    if content_features.dim() != 3 or content_features.shape[-1] != _ops.HEAD_DIM:
        raise ValueError("adain expects head-split features of shape (B*H, L, 64)")
    bh, length, d = content_features.shape
    x = content_features.reshape(bh, 1, length, d)
    mean, std = _ops.token_stats(x, heads=1)                      # (BH,1,1,64) fp32
    a = style_std.reshape(bh, 1, 1, d).float() / (std + _ops.ADAIN_EPS)
    b = style_mean.reshape(bh, 1, 1, d).float() - mean * a
    return _ops.adain_apply(x, a.contiguous(), b.contiguous(), heads=1).reshape(bh, length, d)


# ------------------------------------------------------------------------------------------
# shared prologue / epilogue of every processor (attn_processors.py:42-70, 84-97)
# ------------------------------------------------------------------------------------------
class _Prepared:
    __slots__ = ("hidden", "encoder", "residual", "ndim", "shape4")


def _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb) -> _Prepared:
    st = _Prepared()
    st.residual = hidden_states
    if attn.spatial_norm is not None:
        hidden_states = attn.spatial_norm(hidden_states, temb)
    st.ndim = hidden_states.ndim
    st.shape4 = None
    if st.ndim == 4:
        st.shape4 = hidden_states.shape
        bsz, ch, hh, ww = st.shape4
        hidden_states = hidden_states.view(bsz, ch, hh * ww).transpose(1, 2)
    ref = hidden_states if encoder_hidden_states is None else encoder_hidden_states
    if attn.prepare_attention_mask(attention_mask, ref.shape[1], ref.shape[0]) is not None:
        raise NotImplementedError("attention masks do not occur on the InstantRestore path")
    if attn.group_norm is not None:
        hidden_states = attn.group_norm(hidden_states.transpose(1, 2)).transpose(1, 2)
    st.hidden = hidden_states
    st.encoder = encoder_hidden_states
    return st


def _kv_source(attn, st: _Prepared) -> torch.Tensor:
    if st.encoder is None:
        return st.hidden
    if attn.norm_cross:
        return attn.norm_encoder_hidden_states(st.encoder)
    return st.encoder


def _epilogue(attn, st: _Prepared, tokens: torch.Tensor) -> torch.Tensor:
    out = _project_out(attn, tokens)   # linear proj (LoRA-wrapped on the main UNet)
    out = attn.to_out[1](out)      # dropout (p = 0)
    if st.ndim == 4:
        bsz, ch, hh, ww = st.shape4
        out = out.transpose(-1, -2).reshape(bsz, ch, hh, ww)
    if attn.residual_connection:
        out = out + st.residual
    if attn.rescale_output_factor != 1.0:  # x / 1.0 == x bit-for-bit: skip the launch (always 1.0 in SD-Turbo)
        out = out / attn.rescale_output_factor
    return out


def _autocast_or(weight: torch.Tensor, x: torch.Tensor) -> torch.dtype:
    if x.is_cuda and torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return weight.dtype


def _own_gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    """every projection shape of the SD-Turbo topology (K % 64 == 0, N % 64 == 0) runs this library's GEMMs
    (``ir_linear_fwd``: X-stationary kernels for large M at K <= 320 / K = 640, the LDS-tiled kernel for K = 1280 and
    the small-M shapes); anything else stays ``F.linear``"""
    return bool(_ops.linear_supported(x, w, bias))


def _linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    if _ops.linear_supported(x, w, bias):
        return _ops.linear(x, w, bias)
    return torch.nn.functional.linear(x, w, bias)


LOG2E = 1.4426950408889634
FUSED_CAST = True   # own GEMM: fp32 activations (autocast) are rounded to the compute dtype while they are loaded
PRESCALE_Q = True   # own fused q/k/v GEMM: q leaves its epilogue as Q * attn.scale * log2(e) (still one rounding)


FOLD_MIN_REF_TOKENS = 8   # reference token axes shorter than this get their AdaIN applied to V (one pass) instead of folded into the kernel
FUSED_STATS = True  # own fused q/k/v GEMM: the token statistics of its V third (AdaIN) are its workgroups' tail, no pass over V


def _project_qkv(attn, st: _Prepared, want_stats: bool = False):
    """``to_q`` / ``to_k`` / ``to_v`` of the reference (attn_processors.py:222-230).  Returns
    ``(q, k, v, q_prescaled, v_stats)``; ``v_stats`` is ``None`` unless ``want_stats`` and the fused GEMM could leave
    the partial token statistics of V behind (``ops.ColumnStats``: what ``adain`` needs of V, attn_processors.py:9-10,
    :244-245, without a pass over it).

    Self-attention whose three projections are bias-free linear maps of equal shape - plain
    ``nn.Linear`` or peft LoRA wrappers in inference state (``lora_fold``) - runs them as ONE
    GEMM against a cached concatenation of the three (LoRA-folded) weights (SURVEY.md section
    8f rank 4): the fused attention kernel reads q / k / v as strided views of that single
    ``(B, L, 3C)`` result, so nothing is copied.  Same values as three separate calls for plain
    projections (each output column is the same dot product); for LoRA wrappers the adapter is
    merged exactly as peft's ``merge_and_unload`` does.  Anything else - cross attention, biased
    projections, active dropout, DoRA, training - takes the three module calls like the
    reference.

    Pre-scaled Q: when the fused GEMM is this library's own kernel (``ir_linear_fwd_scaled``), the q third of its
    output is multiplied by ``attn.scale * log2(e)`` in the fp32 accumulator, before the one rounding to 16 bit every
    projection output gets anyway.  The attention kernel is told so (``IR_FLAG_Q_PRESCALED``) and drops the
    multiply-add per score.  Shapes that go to the vendor GEMM keep the plain q."""
    src = _kv_source(attn, st)
    tq, tk, tv = attn.to_q, attn.to_k, attn.to_v
    if torch.is_grad_enabled():
        return tq(st.hidden), tk(src), tv(src), False, None
    if st.encoder is not None:
        # cross attention (the f-1 layers, attn_processors.py:224-230 with encoder_hidden_states): q from the image tokens,
        # k / v from the text states - different inputs, so no q/k/v fusion, but still this library's GEMMs (round 4: the
        # tiny-M shapes lead the vendor GEMM + its cast pass, profiles/r4_gemm_probe_small.txt) and ONE GEMM for k and v
        q = _project_single(attn, "_ir_q_cache", "_ir_q_bias_cache", tq, st.hidden)
        ek, ev = _lora.effective_linear(tk), _lora.effective_linear(tv)
        if ek is not None and ev is not None and ek[0].bias is None and ev[0].bias is None and ek[0].weight.shape == ev[0].weight.shape:
            dtype = _autocast_or(ek[0].weight, src)
            w = _lora.cached_weight(attn, "_ir_kv_cache", (tk, tv), dtype)
            xs = src
            if xs.dtype != dtype and not (FUSED_CAST and _own_gemm(xs, w, None)):
                xs = xs.to(dtype)
            kv = _linear(xs, w, None)
            c = w.shape[0] // 2
            return q, kv[..., :c], kv[..., c:], False, None
        return q, tk(src), tv(src), False, None
    effs = [_lora.effective_linear(m) for m in (tq, tk, tv)]
    if any(e is None or e[0].bias is not None for e in effs) or \
            not (effs[0][0].weight.shape == effs[1][0].weight.shape == effs[2][0].weight.shape):
        return tq(st.hidden), tk(src), tv(src), False, None
    dtype = _autocast_or(effs[0][0].weight, st.hidden)
    w = _lora.cached_weight(attn, "_ir_qkv_cache", (tq, tk, tv), dtype)
    c = w.shape[0] // 3
    x = st.hidden
    own = _own_gemm(x, w, None)      # fp32 activations go straight in: the own GEMM casts them while loading
    if x.dtype != dtype and not (own and FUSED_CAST):
        x = x.to(dtype)
    presc = bool(PRESCALE_Q and c % 32 == 0 and own and _ops.tuning_supports_prescaled_q())
    kw = dict(scale_cols=c, col_scale=float(attn.scale) * LOG2E) if presc else {}
    vstats = None
    if want_stats and FUSED_STATS and own and x.dim() == 3 and c % _ops.HEAD_DIM == 0:
        # token statistics of the V third as the GEMM's tail: whole row blocks per token set (rows | L), whole heads
        rows = _ops.linear_stats_rows(x.shape[0] * x.shape[1], 3 * c, w.shape[1], False)
        if rows > 0 and x.shape[1] % rows == 0 and x.shape[1] // rows <= _ops.STATS_MAX_CHUNKS:
            qkv, vstats = _ops.linear(x, w, None, stats=(2 * c, c), **kw)
    if vstats is None:
        qkv = _ops.linear(x, w, None, **kw) if presc else _linear(x, w, None)
    return qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:], presc, vstats


def _project_single(attn, slot: str, bias_slot: str, module, x: torch.Tensor) -> torch.Tensor:
    """one projection module (plain ``nn.Linear`` or a foldable LoRA wrapper in inference state) as one GEMM of this library
    against its cached folded weight; the module call itself when it cannot be folded"""
    eff = _lora.effective_linear(module)
    if eff is None:
        return module(x)
    dtype = _autocast_or(eff[0].weight, x)
    w = _lora.cached_weight(attn, slot, (module,), dtype)
    bias = eff[0].bias
    if bias is not None and bias.dtype != dtype:
        bias = _lora.cached_cast(attn, bias_slot, bias, dtype)
    if x.dtype != dtype and not (FUSED_CAST and _own_gemm(x, w, bias)):
        x = x.to(dtype)
    return _linear(x, w, bias)


def _project_kv_only(attn, st: _Prepared):
    """``to_k`` / ``to_v`` alone (the last K/V-capture layer under early exit): the lower two thirds of
    the cached fused weight when there is one, the two module calls otherwise."""
    src = _kv_source(attn, st)
    tk, tv = attn.to_k, attn.to_v
    if st.encoder is None and not torch.is_grad_enabled():
        effs = [_lora.effective_linear(m) for m in (attn.to_q, tk, tv)]
        if all(e is not None and e[0].bias is None for e in effs) and \
                effs[0][0].weight.shape == effs[1][0].weight.shape == effs[2][0].weight.shape:
            dtype = _autocast_or(effs[0][0].weight, st.hidden)
            w = _lora.cached_weight(attn, "_ir_qkv_cache", (attn.to_q, tk, tv), dtype)
            c = w.shape[0] // 3
            x = st.hidden
            if x.dtype != dtype and not (FUSED_CAST and _own_gemm(x, w[c:], None)):
                x = x.to(dtype)
            kv = _linear(x, w[c:], None)
            return kv[..., :c], kv[..., c:]
    return tk(src), tv(src)


def _project_out(attn, tokens: torch.Tensor) -> torch.Tensor:
    """``to_out[0]`` (attn_processors.py:267): one GEMM also when the module is a LoRA wrapper in
    inference state; the module call itself in training or when it cannot be folded."""
    proj = attn.to_out[0]
    if torch.is_grad_enabled():
        return proj(tokens)
    eff = _lora.effective_linear(proj)
    if eff is None:
        return proj(tokens)
    dtype = _autocast_or(eff[0].weight, tokens)
    w = _lora.cached_weight(attn, "_ir_out_cache", (proj,), dtype)
    bias = eff[0].bias
    if bias is not None and bias.dtype != dtype:
        bias = _lora.cached_cast(attn, "_ir_out_bias_cache", bias, dtype)
    x = tokens if tokens.dtype == dtype else tokens.to(dtype)
    return _linear(x, w, bias)


def _stats_cached(value: torch.Tensor, cstats, heads: int):
    """AdaIN affine from cached content statistics, reading only V_self (``ir_adain_stats_cached``)"""
    return _ops.adain_stats_cached(value, cstats[0], cstats[1], heads=heads)


def _same_16bit(q: torch.Tensor, *others: Optional[torch.Tensor]) -> List[Optional[torch.Tensor]]:
    """Under autocast the projections emit fp16/bf16; captured reference K/V have that dtype
    too.  Anything else on this path is a caller error worth hearing about."""
    res = []
    for t in others:
        if t is not None and t.dtype != q.dtype:
            raise TypeError(f"shared attention: query is {q.dtype} but a key/value tensor is {t.dtype}")
        res.append(t)
    return res


# ------------------------------------------------------------------------------------------
# K/V-capturing processor of the frozen reference UNet (attn_processors.py:22-97)
# ------------------------------------------------------------------------------------------
def _plain_setattr(self, name, value):
    """``nn.Module.__setattr__`` for modules that own no parameter, buffer or submodule (these processors: strict
    ``load_state_dict`` with an empty state, inference/test.py:47-50): plain attributes - the K/V stash, flags, events, streams -
    skip the registry walk (2 us each, ~170 per eager one-identity step: tools/gpu_host_overhead.py); anything that IS a
    parameter / module, or a name a registry already holds, takes the normal path."""
    if (type(value) in _PLAIN_VALUE_TYPES or not isinstance(value, (nn.Parameter, nn.Module))) and \
            name not in self._parameters and name not in self._buffers and name not in self._modules:
        object.__setattr__(self, name, value)
    else:
        nn.Module.__setattr__(self, name, value)


_PLAIN_VALUE_TYPES = frozenset((type(None), bool, int, float, str, list, tuple, dict, torch.Tensor))   # exact types: no registry business


class ReferenceCaptureComplete(Exception):
    """raised by the LAST K/V-capturing processor when early exit is armed (``kv_harvest``): every
    ``keys`` / ``values`` the main UNet will read is stashed, the rest of the reference UNet's forward
    only produces an output the caller throws away (inference/test.py:100)"""


class AttnProcessor(nn.Module):
    r"""Plain attention that stashes the PRE-head-split ``key`` / ``value`` projections
    ``(B*N, L, C)`` for later sharing (attn_processors.py:73-74)."""

    __setattr__ = _plain_setattr

    def __init__(self):
        super().__init__()
        self.keys, self.values = None, None
        self.is_self_attn = None
        self.stop_after_capture = None    # plain attribute (not a parameter / buffer): kv_harvest arms it with
                                          # the list of all capturing processors of the UNet
        self.record_events = False        # True: record `ready` on the current stream once K/V are stashed
        self.ready = None
        self.stream = None                # HIP stream the K/V were produced on (kv_harvest orders its zero fill after it)
        self.capture_stats = False        # True (kv_harvest.enable_ref_stats): also stash the AdaIN CONTENT statistics of
        self.v_mean, self.v_std = None, None   # every reference V, fp32 (B*N, H, 64) - constant per identity
        self.v_part = None                # ... or (round 4) the partials the q/k/v GEMM left behind (ops.ColumnStats): merged
                                          # by the shared layer's affine kernel, or into (mean, std) on demand

    def reset(self):
        self.keys, self.values = None, None
        self.is_self_attn = None
        self.ready = None
        self.stream = None
        self.v_mean, self.v_std = None, None
        self.v_part = None

    def _stash_stats(self, attn, vstats=None):
        """mean and unbiased std over the tokens of every captured V, per (head, channel): the content statistics of
        ``adain`` (attn_processors.py:9-10) computed HERE, once per reference and on the capture stream, instead of in
        every shared layer of every frame.  ``vstats``: the partials the q/k/v GEMM left behind (round 4: no pass over V,
        one 64-thread-per-(set, head) merge); without them ``ir_token_stats`` reads V."""
        self.v_mean, self.v_std, self.v_part = None, None, None     # never hand a later harvest the statistics of an earlier capture
        if self.capture_stats and self.values.is_cuda:
            if vstats is not None:
                self.v_part = vstats       # merged where they are consumed (kv_harvest / SharedAttnProcessor): no launch here
            else:
                m, sd = _ops.token_stats(self.values.unsqueeze(1), heads=attn.heads)      # (B*N, 1, H, 64) each
                self.v_mean, self.v_std = m[:, 0], sd[:, 0]

    def _mark_ready(self):
        if self.keys.is_cuda:
            self.stream = torch.cuda.current_stream(self.keys.device)
            if self.record_events:
                self.ready = torch.cuda.Event()
                self.ready.record(self.stream)

    def forward(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None):
        st = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        self.is_self_attn = encoder_hidden_states is None
        group = self.stop_after_capture
        if group and all(p.keys is not None for p in group if p is not self):
            # every other capturing layer has run: this is the last one, and only its to_k / to_v are still
            # needed - no query, no attention, no out projection
            self.keys, self.values = _project_kv_only(attn, st)
            self._stash_stats(attn)
            self._mark_ready()
            raise ReferenceCaptureComplete()
        query, key, value, presc, vstats = _project_qkv(attn, st, want_stats=bool(self.capture_stats))
        self.keys, self.values = key, value  # consumed in place by the shared layers: no copies
        self._stash_stats(attn, vstats)
        self._mark_ready()
        _same_16bit(query, key, value)
        kw = {"q_prescaled": True} if presc else {}
        tokens = _ops.shared_attention(query, key, value, heads=attn.heads, scale=attn.scale, include_self=True, **kw)
        return _epilogue(attn, st, tokens)


# ------------------------------------------------------------------------------------------
# face-embedding cross attention (attn_processors.py:100-180); off by default in the configs
# ------------------------------------------------------------------------------------------
class FaceIDAttnProcessor(nn.Module):
    def __init__(self, hidden_size, self_attn_idx=None, cross_attention_dim=None, embed_dim: int = 512):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        width = cross_attention_dim or hidden_size
        self.face_projection = nn.Linear(embed_dim, width)
        self.to_k_face_embed = nn.Linear(width, hidden_size, bias=False)
        self.to_v_face_embed = nn.Linear(width, hidden_size, bias=False)
        self.self_attn_idx = self_attn_idx
        self.keys, self.values = None, None
        self.is_self_attn = None

    def reset(self):
        self.keys, self.values = None, None
        self.is_self_attn = None

    def forward(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                ref_keys=None, ref_values=None, ref_events=None, ref_stats=None, ref_valid=None):
        # ref_* accepted and ignored, like the reference: the host forwards ONE cross_attention_kwargs dict to every processor
        # (unet.py / diffusers), so whatever the harvest hands SharedAttnProcessor (ref_valid included) arrives here too
        st = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        self.is_self_attn = encoder_hidden_states is None
        query = attn.to_q(st.hidden)
        src = self.face_projection(_kv_source(attn, st))
        key, value = self.to_k_face_embed(src), self.to_v_face_embed(src)
        _same_16bit(query, key, value)
        tokens = _ops.shared_attention(query, key, value, heads=attn.heads, scale=attn.scale, include_self=True)
        return _epilogue(attn, st, tokens)


# ------------------------------------------------------------------------------------------
# the hot path: shared-image attention (attn_processors.py:183-279)
# ------------------------------------------------------------------------------------------
class SharedAttnProcessor(nn.Module):
    r"""Extended self-attention over ``[self K/V (iff train_input)] ++ N x reference K/V``.

    ``ref_keys[self_attn_idx]`` / ``ref_values[...]`` are the ``(B, N, L, C)`` tensors produced
    by ``get_conditioning_keys_values`` (pix2pix_turbo.py:265-266); they are read in place by the
    kernel (segment walk) - N is taken from the tensor, zero-filled references keep their
    ``exp(0)`` weight, and with ``use_adain`` every reference V is renormalised to the
    statistics of this image's own V inside the kernel's V staging.
    """

    __setattr__ = _plain_setattr

    def __init__(self, self_attn_idx: int = None, save_self_attentions: bool = False,
                 use_adain: bool = False, train_input: bool = True):
        super().__init__()
        self.self_attn_idx = self_attn_idx
        self.save_self_attentions = save_self_attentions
        self.use_adain = use_adain
        self.train_input = train_input
        # opt-in (round 5; a plain attribute like ``AttnProcessor.record_events``, the constructor stays the reference's): when set,
        # every call also leaves ``attention_mass`` - fp32 (B, H, L, [self?] + N), the attention mass per K/V segment - which is what
        # gradio_demo.py:119-127 reduces ``attention_probs`` to (``probs[..., attn_size*idx : attn_size*(idx+1)].sum(-1)``), without
        # the (B, H, L, Lkv) tensor and without a second pass (``ir_shared_attn_args.seg_mass``, ABI v9).  Independent of ``save_self_attentions``, whose meaning is unchanged.
        # COLUMN ORDER vs the reference's demo: gradio_demo.py slices blocks ``attn_size*idx`` for idx 0..3 starting at column 0.  With
        # ``train_input`` the columns are [self, ref0, ref1, ...], so the demo's block 0 is the SELF segment and its last reference is
        # never read.  Reference-parity ranking therefore uses ``attention_mass[..., 0:4]`` (meaningful only when Ls == Lr, which the
        # model guarantees); ``attention_mass[..., int(train_input):]`` is the per-REFERENCE form (what the demo presumably meant).
        self.save_attention_mass = False
        self.attention_mass = None

    def forward(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                ref_keys=None, ref_values=None, ref_events=None, ref_stats=None, ref_valid=None):
        """``ref_valid`` (optional, round 5): int32 ``(B,)`` device tensor from the harvest (``kv_harvest`` ``with_valid``) when
        it zero-filled references ``n >= valid[b]`` (pix2pix_turbo.py:269-273): the kernel then closes those all-zero segments
        analytically instead of walking them (ABI v8 ``valid_refs``).  Same output; the tokens keep their exp(0) weight."""
        st = _prologue(attn, hidden_states, encoder_hidden_states, attention_mask, temb)
        shared = self.self_attn_idx is not None and ref_keys is not None and ref_values is not None
        # the GEMM's statistics tail (style partials of this layer's own V) only pays when the content statistics arrive
        # precomputed (``ref_stats``): without them ir_adain_stats reads every V anyway and would throw the partials away
        have_cstats = bool(shared and ref_stats is not None and ref_stats[self.self_attn_idx] is not None)
        query, key, value, presc, vstats = _project_qkv(attn, st, want_stats=bool(self.use_adain and have_cstats))

        ref_k = ref_v = None
        include_self = True
        affine = None
        if shared:
            ref_k = ref_keys[self.self_attn_idx]
            ref_v = ref_values[self.self_attn_idx]
            cstats = ref_stats[self.self_attn_idx] if ref_stats is not None else None
            if ref_events is not None and ref_events[self.self_attn_idx] is not None:
                # the reference UNet runs on another HIP stream (kv_harvest with_events): this layer needs
                # capture layer `self_attn_idx` and nothing later
                cur = torch.cuda.current_stream(ref_k.device)
                cur.wait_event(ref_events[self.self_attn_idx])
                ref_k.record_stream(cur)
                ref_v.record_stream(cur)
                if hasattr(cstats, "finished"):
                    cstats.record_stream(cur)
                elif cstats is not None:
                    cstats[0].record_stream(cur)
                    cstats[1].record_stream(cur)
            elif hasattr(cstats, "sync_to_current"):
                cstats.sync_to_current()      # no event handed over: order this stream behind the partials' producer (no-op on one stream)
            include_self = bool(self.train_input)
            if self.use_adain:
                # style = this image's own post-projection V; content = each reference V.  With the content statistics
                # handed over (``ref_stats``: computed once per identity by the K/V-capture layer, kv_harvest with_stats)
                # only V_self is read here - 1/(N+1) of the bytes, the same (a, b) bit for bit
                # Round 4: the style statistics arrive as the partials this layer's own q/k/v GEMM left behind (``vstats``):
                # with them and the content statistics nothing of V is read here at all - one small launch
                if hasattr(cstats, "finished"):
                    if vstats is not None:    # both sides as partials: ONE launch merges them into the affine
                        affine = _ops.adain_affine_from_partials(vstats, value.shape[0], value.shape[1], ref_v.shape[1], ref_v.shape[2],
                                                                 content=cstats.part, valid=cstats.valid)
                    else:
                        affine = _stats_cached(value, cstats.finished(), attn.heads)
                elif cstats is not None and vstats is not None:
                    affine = _ops.adain_affine_from_partials(vstats, value.shape[0], value.shape[1], ref_v.shape[1], ref_v.shape[2],
                                                             content_mean=cstats[0], content_std=cstats[1])
                elif cstats is not None:
                    affine = _stats_cached(value, cstats, attn.heads)
                else:
                    affine = _ops.adain_stats(value, ref_v, heads=attn.heads)
        _same_16bit(query, key, value, ref_k, ref_v)
        if affine is not None and ref_v.shape[2] < FOLD_MIN_REF_TOKENS:
            # a handful of reference tokens: a channel whose few values lie ulps apart gets a ratio in the thousands, and the
            # fold's a * sum(p~ v) + b * sum(p) would amplify the 16-bit rounding of P by a * |mean|.  Here the
            # renormalised V is formed once (ir_adain_apply, fp32 arithmetic, one rounding - what the reference's adain()
            # does, attn_processors.py:7-18) and the attention runs without an affine.  Never taken by the model's own axes (>= 256).
            ref_v = _ops.adain_apply(ref_v, affine[0], affine[1], heads=attn.heads)
            affine = None
            ref_valid = None       # a zero-filled reference's V is the style mean now, not zero: its tiles are walked

        want_probs = bool(self.save_self_attentions)
        want_mass = bool(getattr(self, "save_attention_mass", False))
        kw = {"q_prescaled": True} if presc else {}
        if shared and ref_valid is not None:
            kw["valid_refs"] = ref_valid
        # the masses are a by-product of the attention launch itself (ABI v9 ``seg_mass``: the kernels hold the row sums at every
        # segment boundary): no second pass over Q and K
        res = _ops.shared_attention(query, key, value, ref_k, ref_v, heads=attn.heads, scale=attn.scale,
                                    include_self=include_self, adain=affine, return_lse=want_probs, return_mass=want_mass, **kw)
        if want_mass:
            self.attention_mass = res[-1]
            res = res[:-1] if want_probs else res[0]
        if want_probs:
            tokens, lse = res
            # (B, H, L, Lkv), columns [self?] ++ ref0 ++ ... ++ refN-1, in the compute dtype.  A pre-scaled query
            # already carries scale * log2(e): its scores are exponents, ln 2 turns them into the logits of the LSE
            self.attention_probs = _ops.attn_probs(query, key, ref_k, lse, heads=attn.heads,
                                                   scale=0.6931471805599453 if presc else attn.scale,
                                                   include_self=include_self)
        else:
            tokens = res
        return _epilogue(attn, st, tokens)


# ------------------------------------------------------------------------------------------
# registration: the plugin boundary (attn_processors.py:282-331)
# ------------------------------------------------------------------------------------------
def _hidden_size_for(name: str, block_out_channels) -> Optional[int]:
    if name.startswith("mid_block"):
        return block_out_channels[-1]
    if name.startswith("up_blocks"):
        return list(reversed(block_out_channels))[int(name[len("up_blocks.")])]
    if name.startswith("down_blocks"):
        return block_out_channels[int(name[len("down_blocks.")])]
    return None


def register_attention_processor(unet, cfg, save_self_attentions: bool = False):
    """Install a :class:`SharedAttnProcessor` on every attention of the main UNet.

    Walks ``unet.attn_processors`` in registration order; the ``up_blocks.*attn1`` layers get
    ``self_attn_idx`` 0..8 (``up_blocks.1.attentions.{0,1,2}``, then ``.2``, then ``.3``); all
    other self-attentions and every cross-attention get ``self_attn_idx=None`` (cross-attention
    becomes :class:`FaceIDAttnProcessor` when ``cfg.condition_on_face_embeds``).  ``cfg`` is the
    reference's ``ModelConfig`` or anything with ``use_adain``, ``train_input`` and
    ``condition_on_face_embeds`` attributes.
    """
    procs = {}
    next_idx = 0
    face_ids = bool(getattr(cfg, "condition_on_face_embeds", False))
    for name in unet.attn_processors.keys():
        is_cross = not name.endswith("attn1.processor")
        if is_cross:
            if face_ids:
                proc = FaceIDAttnProcessor(hidden_size=_hidden_size_for(name, unet.config.block_out_channels),
                                           self_attn_idx=None,
                                           cross_attention_dim=unet.config.cross_attention_dim, embed_dim=512)
            else:
                proc = SharedAttnProcessor(self_attn_idx=None, use_adain=cfg.use_adain, train_input=cfg.train_input)
        elif name.startswith("up_blocks") and "attn1" in name:
            proc = SharedAttnProcessor(self_attn_idx=next_idx, save_self_attentions=save_self_attentions,
                                       use_adain=cfg.use_adain, train_input=cfg.train_input)
            next_idx += 1
        else:
            proc = SharedAttnProcessor(self_attn_idx=None, save_self_attentions=save_self_attentions,
                                       use_adain=cfg.use_adain, train_input=cfg.train_input)
        procs[name] = proc.to(unet.device, dtype=unet.dtype)
    unet.set_attn_processor(procs)
    _lora.invalidate_all(unet)            # (re-)registration starts from freshly folded projection weights
    _lora.install_invalidation_hook(unet)


def register_attention_processor_kv_unet(unet):
    """Put the K/V-capturing :class:`AttnProcessor` on the decoder self-attentions of the frozen
    reference UNet and leave every other processor as it is (attn_processors.py:324-331)."""
    current = unet.attn_processors
    procs = {}
    for name, proc in current.items():
        if name.startswith("up_blocks") and "attn1" in name:
            procs[name] = AttnProcessor().to(unet.device, dtype=unet.dtype)
        else:
            procs[name] = proc
    unet.set_attn_processor(procs)
    _lora.invalidate_all(unet)
    _lora.install_invalidation_hook(unet)


__all__ = ["adain", "AttnProcessor", "FaceIDAttnProcessor", "SharedAttnProcessor",
           "register_attention_processor", "register_attention_processor_kv_unet"]
