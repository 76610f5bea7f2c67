"""Tensor-level entry points of the HIP hot path.

Thin, allocation-only wrappers: they check devices / dtypes / strides, allocate outputs with
torch (device memory and streams are PyTorch-ROCm plumbing) and hand raw pointers, element
strides and the current HIP stream to the C ABI (``include/instantrestore_hip.h``).  Nothing
here computes anything, and nothing falls back to torch math: a CPU tensor or a missing
library is an error.

Layouts are the reference's own, consumed IN PLACE (no head-split copies, no ``cat``):
``q, k_self, v_self``: ``(B, L, H*64)``; ``ref_k, ref_v``: ``(B, N, Lr, H*64)`` exactly as
``Pix2Pix_Turbo.get_conditioning_keys_values`` hands them over (pix2pix_turbo.py:265-266).
"""
from __future__ import annotations

import ctypes as C
import functools
import os
import collections
import threading
from typing import Optional, Tuple

import torch

from . import _lib

HEAD_DIM = 64
ADAIN_EPS = 1e-5  # attn_processors.py:10,245

_DT = {torch.float16: _lib.IR_DTYPE_F16, torch.bfloat16: _lib.IR_DTYPE_BF16}
# measurement hook (bench.py): (predicate(q, ref_k, adain) -> bool, list collecting (start, end) HIP events recorded
# on the launch stream around the matching ir_shared_attn_fwd calls) or None
EVENT_SINK = None


def _dtype_code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(
            f"the fused HIP path computes in fp16/bf16 (got {t.dtype}); run the model under "
            "torch.autocast (as face_replace/inference/test.py:83 does) or cast it to bf16/fp16"
        ) from None


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)   # the handle without a torch.cuda.Stream object around it


def _stream() -> int:
    """the calling thread's current HIP stream on the current device, as the integer the C ABI takes.  Through torch's raw
    accessor where this torch has it (what its own compiled kernels use): ``torch.cuda.current_stream()`` builds a Stream object
    per call, ~7 us of the ~30 us a launch costs the host - a fifth of an eager one-identity step (tools/gpu_host_overhead.py)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """HIP launches go to the calling thread's CURRENT device: make that the device of the tensors
    (so a caller holding cuda:1 tensors while cuda:0 is current gets the right device and stream)."""

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        dev = None
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                dev = a.device
                break
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)

    return wrapper


def _need_gpu(*ts: Optional[torch.Tensor]) -> torch.device:
    dev = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "instantrestore_amd ops run on the MI355X only; got a CPU tensor "
                "(there is no CPU fallback - the CPU restatement lives in oracle/ and is test-only)"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
    return dev


def _tok(t: torch.Tensor, heads: int, name: str) -> torch.Tensor:
    """(B, L, H*64) view usable in place; falls back to .contiguous() only for exotic strides."""
    if t.dim() != 3 or t.shape[-1] != heads * HEAD_DIM:
        raise ValueError(f"{name}: expected (B, L, {heads * HEAD_DIM}), got {tuple(t.shape)}")
    if t.stride(-1) != 1 or t.stride(0) % 8 or t.stride(1) % 8 or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _ref(t: torch.Tensor, heads: int, name: str) -> torch.Tensor:
    if t.dim() != 4 or t.shape[-1] != heads * HEAD_DIM:
        raise ValueError(f"{name}: expected (B, N, L, {heads * HEAD_DIM}), got {tuple(t.shape)}")
    if t.stride(-1) != 1 or any(t.stride(i) % 8 for i in range(3)) or t.data_ptr() % 16:
        t = t.contiguous()
    return t


def _fill_args(q, k_self, v_self, ref_k, ref_v, heads, scale, include_self, adain, out, lse, split=True,
               q_prescaled=False, valid_refs=None):
    a = _lib.SharedAttnArgs()
    a.struct_size = C.sizeof(_lib.SharedAttnArgs)
    a.dtype = _dtype_code(q)
    a.flags = (_lib.IR_FLAG_INCLUDE_SELF if include_self else 0) | (_lib.IR_FLAG_Q_PRESCALED if q_prescaled else 0)
    if out is not None and out.dtype == torch.float32:
        a.flags |= _lib.IR_FLAG_OUT_F32
    a.tuning = _TUNING
    a.batch, a.len_q, _ = q.shape
    a.heads = heads
    a.scale = float(scale)
    a.q = q.data_ptr()
    a.q_sb, a.q_sl, a.q_sh = q.stride(0), q.stride(1), HEAD_DIM
    if include_self:
        a.len_self = k_self.shape[1]
        a.k_self, a.v_self = k_self.data_ptr(), v_self.data_ptr()
        a.ks_sb, a.ks_sl, a.ks_sh = k_self.stride(0), k_self.stride(1), HEAD_DIM
        a.vs_sb, a.vs_sl, a.vs_sh = v_self.stride(0), v_self.stride(1), HEAD_DIM
    if ref_k is not None:
        a.n_refs, a.len_ref = ref_k.shape[1], ref_k.shape[2]
        a.k_ref, a.v_ref = ref_k.data_ptr(), ref_v.data_ptr()
        a.kr_sb, a.kr_sn, a.kr_sl, a.kr_sh = ref_k.stride(0), ref_k.stride(1), ref_k.stride(2), HEAD_DIM
        a.vr_sb, a.vr_sn, a.vr_sl, a.vr_sh = ref_v.stride(0), ref_v.stride(1), ref_v.stride(2), HEAD_DIM
    if adain is not None:
        a.adain_a, a.adain_b = adain[0].data_ptr(), adain[1].data_ptr()
    if valid_refs is not None and ref_k is not None:
        if valid_refs.dtype != torch.int32 or valid_refs.device != q.device or valid_refs.numel() != q.shape[0] or not valid_refs.is_contiguous():
            raise ValueError(f"valid_refs must be a contiguous int32 ({q.shape[0]},) tensor on {q.device}")
        a.valid_refs = valid_refs.data_ptr()
    if out is not None:
        a.out = out.data_ptr()
        a.o_sb, a.o_sl, a.o_sh = out.stride(0), out.stride(1), HEAD_DIM
    if lse is not None:
        a.lse = lse.data_ptr()
    if out is not None and split:  # scratch for the remainder split (ir_shared_attn_workspace_bytes)
        ws = _workspace(q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        a._keepalive = ws
    return a


_WS = threading.local()
_WS_MAX_PER_THREAD = 8   # unpinned (device, stream) pairs a thread keeps scratch for; least recently used goes first


def _workspace(device: torch.device) -> torch.Tensor:
    """one scratch buffer per (device, stream, THREAD).  Launches on one stream are ordered, so one thread reuses its buffer
    safely; but ``ir_shared_attn_fwd`` issues two launches per call (the kernel, then the merge of its K/V-range pieces through
    this scratch), ctypes releases the GIL for the call, and threads that share a stream - torch's default stream is shared by
    all threads - can interleave: A's kernel, B's kernel, A's merge would read B's pieces.  A buffer per thread removes the
    hazard (69 MB each; tests/test_gpu_threads.py).

    Bounded (round 6): the buffers live in ``threading.local()`` - they are released with their thread, so a server that
    spawns a thread per request does not accumulate them - and a thread keeps at most ``_WS_MAX_PER_THREAD`` UNPINNED
    (device, stream) pairs, least recently used evicted.  A buffer handed out while its stream is being CAPTURED into a hipGraph
    is pinned for the life of the thread: the graph replays with the raw pointer, long after this call (bench.py's captured step;
    the first version of the bound evicted such a buffer and the replay faulted).  An evicted buffer goes back to torch's caching
    allocator, which keeps it off other streams until the work queued on its stream has run (``record_stream``)."""
    key = (device.index, _RAW_STREAM(device.index) if _RAW_STREAM is not None and device.index is not None
           else torch.cuda.current_stream(device).cuda_stream)
    cache = getattr(_WS, "cache", None)
    if cache is None:
        cache = _WS.cache = collections.OrderedDict()
        _WS.pinned = {}
    ws = _WS.pinned.get(key)
    if ws is not None:
        return ws
    capturing = torch.cuda.is_current_stream_capturing()
    ws = cache.get(key)
    if ws is None:
        ws = torch.empty(_lib.lib().ir_shared_attn_workspace_bytes() // 4, dtype=torch.float32, device=device)
        if not capturing:
            ws.record_stream(torch.cuda.current_stream(device))
        cache[key] = ws
    else:
        cache.move_to_end(key)
    if capturing:
        _WS.pinned[key] = cache.pop(key)
    while len(cache) > _WS_MAX_PER_THREAD:
        cache.popitem(last=False)
    return ws


def _forward_only(*ts) -> None:
    """the HIP path is the inference forward (test.py runs under ``torch.no_grad()``): a tensor that would
    need a backward through it must not pass silently - the result would carry no ``grad_fn``"""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts):
        raise NotImplementedError(
            "instantrestore_amd ops are forward-only (inference hot path): run under torch.no_grad() / "
            "inference_mode(); the training backward of the reference (coach.py) is out of scope")


def _prep(q, k_self, v_self, ref_k, ref_v, heads, include_self, adain):
    _need_gpu(q, k_self, v_self, ref_k, ref_v)
    _forward_only(q, k_self, v_self, ref_k, ref_v)
    q = _tok(q, heads, "q")
    if (ref_k is None) != (ref_v is None):
        raise ValueError("ref_k and ref_v must be given together")
    if ref_k is None and not include_self:
        raise ValueError("no references and include_self=False: empty key/value sequence")
    if include_self:
        k_self, v_self = _tok(k_self, heads, "k_self"), _tok(v_self, heads, "v_self")
        if k_self.shape != v_self.shape or k_self.shape[0] != q.shape[0]:
            raise ValueError("k_self / v_self shape mismatch")
    if ref_k is not None:
        ref_k, ref_v = _ref(ref_k, heads, "ref_k"), _ref(ref_v, heads, "ref_v")
        if ref_k.shape != ref_v.shape or ref_k.shape[0] != q.shape[0]:
            raise ValueError("ref_k / ref_v shape mismatch")
    for t in (k_self if include_self else None, v_self if include_self else None, ref_k, ref_v):
        if t is not None and t.dtype != q.dtype:
            raise TypeError(f"mixed dtypes on the fused path: q is {q.dtype}, a K/V tensor is {t.dtype}")
    if adain is not None:
        if ref_k is None:
            raise ValueError("AdaIN affine given without references")
        B, N = ref_k.shape[:2]
        for t in adain:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != B * N * heads * HEAD_DIM:
                raise ValueError("adain (a, b) must be contiguous fp32 of shape (B, N, H, 64)")
    return q, k_self, v_self, ref_k, ref_v


@_on_tensor_device
def shared_attention(q, k_self, v_self, ref_k=None, ref_v=None, *, heads: int, scale: float,
                     include_self: bool = True, adain: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                     return_lse: bool = False, split: bool = True, q_prescaled: bool = False,
                     out_dtype: Optional[torch.dtype] = None, valid_refs: Optional[torch.Tensor] = None,
                     return_mass: bool = False):
    """Fused extended self-attention (``ir_shared_attn_fwd``).

    Returns ``out`` (B, Lq, H*64) in q's dtype [, ``lse`` (B, H, Lq) fp32] [, ``mass`` (B, H, Lq, include_self + N) fp32:
    ``return_mass``, ABI v9 ``seg_mass`` - the attention mass of every K/V segment as a by-product of the same launch, what
    gradio_demo.py:119-127 reduces ``attention_probs`` to].  ``adain`` is the
    (a, b) pair from :func:`adain_stats`; the reference-V renormalisation happens inside the
    kernel's V staging.  ``q_prescaled``: ``q`` already holds ``Q * scale * log2(e)`` (``IR_FLAG_Q_PRESCALED``: the
    fused q/k/v projection folds the factor into its weights; ``scale`` stays the reference's ``attn.scale``).
    ``out_dtype=torch.float32`` returns the result before its rounding to 16 bit (``IR_FLAG_OUT_F32``, parity tests).
    ``valid_refs`` (ABI v8): int32 ``(B,)`` on the device - a PROMISE that references ``n >= valid_refs[b]`` are all-zero in
    ``ref_k`` and ``ref_v`` (``zero_invalid_refs``; pix2pix_turbo.py:269-273).  The kernels then close that suffix of the reference
    list analytically (every score exactly 0, every value row 0 or the AdaIN shift) instead of walking its tiles: same result,
    time proportional to the valid segments.  Zeroed, not masked: the tokens keep their exp(0) weight.
    """
    q, k_self, v_self, ref_k, ref_v = _prep(q, k_self, v_self, ref_k, ref_v, heads, include_self, adain)
    if out_dtype not in (None, q.dtype, torch.float32):
        raise TypeError("out_dtype must be the compute dtype or torch.float32")
    out = torch.empty((q.shape[0], q.shape[1], heads * HEAD_DIM), dtype=out_dtype or q.dtype, device=q.device)
    lse = torch.empty((q.shape[0], heads, q.shape[1]), dtype=torch.float32, device=q.device) if return_lse else None
    args = _fill_args(q, k_self, v_self, ref_k, ref_v, heads, scale, include_self, adain, out, lse, split, q_prescaled, valid_refs)
    mass = None
    if return_mass:
        nseg = (1 if include_self else 0) + (ref_k.shape[1] if ref_k is not None else 0)
        mass = torch.empty((q.shape[0], heads, q.shape[1], nseg), dtype=torch.float32, device=q.device)
        args.seg_mass = mass.data_ptr()
    sink = EVENT_SINK
    if sink is not None and sink[0](q, ref_k, adain):   # bench.py: HIP events around chosen launches, in situ
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().ir_shared_attn_fwd(C.byref(args), _stream()), "ir_shared_attn_fwd")
        e1.record()
        sink[1].append((e0, e1))
    else:
        _lib.check(_lib.lib().ir_shared_attn_fwd(C.byref(args), _stream()), "ir_shared_attn_fwd")
    res = (out,) + ((lse,) if return_lse else ()) + ((mass,) if return_mass else ())
    return res if len(res) > 1 else out


@_on_tensor_device
def time_shared_attention(q, k_self, v_self, ref_k=None, ref_v=None, *, heads: int, scale: float,
                          include_self: bool = True, adain=None, iters: int = 10, split: bool = True,
                          q_prescaled: bool = False) -> float:
    """Average ms per launch measured with HIP events on the launch stream (``bench.py``)."""
    q, k_self, v_self, ref_k, ref_v = _prep(q, k_self, v_self, ref_k, ref_v, heads, include_self, adain)
    out = torch.empty((q.shape[0], q.shape[1], heads * HEAD_DIM), dtype=q.dtype, device=q.device)
    args = _fill_args(q, k_self, v_self, ref_k, ref_v, heads, scale, include_self, adain, out, None, split, q_prescaled)
    ms = C.c_float(0.0)
    _lib.check(_lib.lib().ir_time_shared_attn_fwd(C.byref(args), int(iters), _stream(), C.byref(ms)),
               "ir_time_shared_attn_fwd")
    return float(ms.value)


def bench_mfma_stream(dtype: torch.dtype = torch.bfloat16, zero_operands: bool = False, iters: int = 40000, launches: int = 5,
                      device: Optional[torch.device] = None) -> float:
    """TFLOP/s an MFMA-only stream of the attention kernels' instruction sustains on this device (``ir_bench_mfma_stream``):
    on pseudo-random operands the board's power cap sets it, on all-zero operands the clock does.  Measurement only."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("bench_mfma_stream needs the GPU (no CPU fallback)")
    with torch.cuda.device(dev):
        nbytes = int(_lib.lib().ir_bench_mfma_stream_scratch_bytes())
        scratch = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        out = C.c_float(0.0)
        _lib.check(_lib.lib().ir_bench_mfma_stream(_DT[dtype], 1 if zero_operands else 0, int(iters), int(launches), scratch.data_ptr(),
                                                   nbytes, _stream(), C.byref(out)), "ir_bench_mfma_stream")
    return float(out.value)


def shared_attention_kernel_name(q, k_self, v_self, ref_k=None, ref_v=None, *, heads: int, scale: float,
                                 include_self: bool = True, adain=None, q_prescaled: bool = False) -> str:
    """which kernel the dispatcher launches for these tensors (reporting only)"""
    q, k_self, v_self, ref_k, ref_v = _prep(q, k_self, v_self, ref_k, ref_v, heads, include_self, adain)
    out = torch.empty((q.shape[0], q.shape[1], heads * HEAD_DIM), dtype=q.dtype, device=q.device)
    args = _fill_args(q, k_self, v_self, ref_k, ref_v, heads, scale, include_self, adain, out, None, True, q_prescaled)
    return _lib.lib().ir_shared_attn_kernel_name(C.byref(args)).decode()


PROBS_KERNELS = {"auto": 0, "generic": 1, "lines64": 2, "lines32": 3, "lines32k128": 4, "lines64k128": 5, "lines32k256": 6}   # IR_PROBS_*


def _probs_args(q, k_self, ref_k, lse, heads, scale, include_self, q_prescaled=False):
    q, k_self, _, ref_k, _ = _prep(q, k_self, k_self, ref_k, ref_k, heads, include_self, None)
    B, Lq, _ = q.shape
    lkv = (k_self.shape[1] if include_self else 0) + (ref_k.shape[1] * ref_k.shape[2] if ref_k is not None else 0)
    if lse.dtype != torch.float32 or not lse.is_contiguous() or tuple(lse.shape) != (B, heads, Lq):
        raise ValueError("lse must be contiguous fp32 (B, H, Lq)")
    args = _fill_args(q, k_self, k_self, ref_k, ref_k, heads, scale, include_self, None, None, lse, True, q_prescaled)
    return args, q, B, Lq, lkv, (q, k_self, ref_k, lse)


@_on_tensor_device
def attn_probs(q, k_self, ref_k, lse, *, heads: int, scale: float, include_self: bool = True, kernel: str = "auto",
               q_prescaled: bool = False) -> torch.Tensor:
    """Materialise ``attention_probs`` (B, H, Lq, Lkv) from the LSE of the fused forward
    (``ir_attn_probs``; the ``save_self_attentions`` dump path, attn_processors.py:258-261).
    ``kernel``: ``PROBS_KERNELS`` (benchmarks / A-B tests; "auto" is what the processors use).  ``q_prescaled``: ``q`` holds
    ``Q * scale * log2(e)`` (``IR_FLAG_Q_PRESCALED``; ``scale`` stays the reference's ``attn.scale``, the unit of ``lse``)."""
    args, q, B, Lq, lkv, _keep = _probs_args(q, k_self, ref_k, lse, heads, scale, include_self, q_prescaled)
    probs = torch.empty((B, heads, Lq, lkv), dtype=q.dtype, device=q.device)
    _lib.check(_lib.lib().ir_attn_probs_ex(C.byref(args), probs.data_ptr(), PROBS_KERNELS[kernel], _stream()), "ir_attn_probs")
    return probs


@_on_tensor_device
def attn_segment_mass(q, k_self, ref_k, lse, *, heads: int, scale: float, include_self: bool = True,
                      q_prescaled: bool = False) -> torch.Tensor:
    """Attention mass per K/V segment, fp32 (B, H, Lq, include_self + N), without the probability matrix
    (``ir_attn_segment_mass``): what gradio_demo.py:119-127 reduces ``attention_probs`` to."""
    args, q, B, Lq, _, _keep = _probs_args(q, k_self, ref_k, lse, heads, scale, include_self, q_prescaled)
    nseg = (1 if include_self else 0) + (ref_k.shape[1] if ref_k is not None else 0)
    mass = torch.empty((B, heads, Lq, nseg), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().ir_attn_segment_mass(C.byref(args), mass.data_ptr(), _stream()), "ir_attn_segment_mass")
    return mass


@_on_tensor_device
def adain_stats(v_self: torch.Tensor, ref_v: torch.Tensor, *, heads: int, eps: float = ADAIN_EPS):
    """AdaIN as a per-(b, n, head, channel) affine ``x*a + b`` (``ir_adain_stats``).

    ``v_self`` (B, L, H*64) supplies the style statistics, ``ref_v`` (B, N, Lr, H*64) the content
    statistics (attn_processors.py:9-10, 244-245: token axis, unbiased std, eps on both)."""
    _need_gpu(v_self, ref_v)
    _forward_only(v_self, ref_v)
    v_self, ref_v = _tok(v_self, heads, "v_self"), _ref(ref_v, heads, "ref_v")
    if v_self.dtype != ref_v.dtype:
        raise TypeError("v_self / ref_v dtype mismatch")
    B, Ls, _ = v_self.shape
    _, N, Lr, _ = ref_v.shape
    L = _lib.lib()
    nbytes = L.ir_adain_stats_workspace_bytes(B, heads, Ls, N, Lr)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=v_self.device)
    a = torch.empty((B, N, heads, HEAD_DIM), dtype=torch.float32, device=v_self.device)
    b = torch.empty_like(a)
    rc = L.ir_adain_stats(_dtype_code(v_self), B, heads, Ls, N, Lr,
                          v_self.data_ptr(), v_self.stride(0), v_self.stride(1), HEAD_DIM,
                          ref_v.data_ptr(), ref_v.stride(0), ref_v.stride(1), ref_v.stride(2), HEAD_DIM,
                          float(eps), a.data_ptr(), b.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "ir_adain_stats")
    return a, b


@_on_tensor_device
def adain_stats_cached(v_self: torch.Tensor, content_mean: torch.Tensor, content_std: torch.Tensor, *, heads: int,
                       eps: float = ADAIN_EPS):
    """The affine of :func:`adain_stats` from CACHED content statistics (``ir_adain_stats_cached``): ``content_mean`` /
    ``content_std`` are the fp32 ``(B, N, H, 64)`` outputs of :func:`token_stats` over the reference V's (the K/V-capture
    layer computes them once per identity); only ``v_self`` is read.  Bit-identical to ``adain_stats(v_self, ref_v)``."""
    _need_gpu(v_self, content_mean, content_std)
    _forward_only(v_self)
    v_self = _tok(v_self, heads, "v_self")
    B, Ls, _ = v_self.shape
    if content_mean.dim() != 4 or content_mean.shape != content_std.shape or content_mean.shape[0] != B or \
            tuple(content_mean.shape[2:]) != (heads, HEAD_DIM):
        raise ValueError(f"content statistics must be two (B, N, {heads}, 64) tensors")
    for t in (content_mean, content_std):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("content statistics must be contiguous fp32")
    N = content_mean.shape[1]
    L = _lib.lib()
    nbytes = L.ir_adain_stats_workspace_bytes(B, heads, Ls, 0, Ls)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=v_self.device)
    a = torch.empty((B, N, heads, HEAD_DIM), dtype=torch.float32, device=v_self.device)
    b = torch.empty_like(a)
    rc = L.ir_adain_stats_cached(_dtype_code(v_self), B, heads, Ls, N, v_self.data_ptr(), v_self.stride(0), v_self.stride(1),
                                 HEAD_DIM, content_mean.data_ptr(), content_std.data_ptr(), float(eps), a.data_ptr(),
                                 b.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "ir_adain_stats_cached")
    return a, b


@_on_tensor_device
def token_stats(x: torch.Tensor, *, heads: int):
    """mean and unbiased std over the token axis of x (B, M, L, H*64) -> two fp32 (B, M, H, 64)
    tensors (``ir_token_stats``)."""
    _need_gpu(x)
    x = _ref(x, heads, "x")
    B, M, Lx, _ = x.shape
    L = _lib.lib()
    nbytes = L.ir_adain_stats_workspace_bytes(B, heads, Lx, M - 1, Lx)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    mean = torch.empty((B, M, heads, HEAD_DIM), dtype=torch.float32, device=x.device)
    std = torch.empty_like(mean)
    rc = L.ir_token_stats(_dtype_code(x), B, heads, M, Lx,
                          x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), HEAD_DIM,
                          mean.data_ptr(), std.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _lib.check(rc, "ir_token_stats")
    return mean, std


@_on_tensor_device
def adain_apply(x: torch.Tensor, a: torch.Tensor, b: torch.Tensor, *, heads: int) -> torch.Tensor:
    """``y = x*a + b`` over (B, N, L, H*64) (``ir_adain_apply``; op-level parity, ``adain()``)."""
    _need_gpu(x, a, b)
    x = _ref(x, heads, "x")
    B, N, Lx, _ = x.shape
    y = torch.empty_like(x, memory_format=torch.contiguous_format)
    rc = _lib.lib().ir_adain_apply(_dtype_code(x), B, heads, N, Lx,
                                   x.data_ptr(), x.stride(0), x.stride(1), x.stride(2), HEAD_DIM,
                                   a.data_ptr(), b.data_ptr(),
                                   y.data_ptr(), y.stride(0), y.stride(1), y.stride(2), HEAD_DIM, _stream())
    _lib.check(rc, "ir_adain_apply")
    return y


@_on_tensor_device
def zero_invalid_refs(k: torch.Tensor, v: torch.Tensor, valid_indices: torch.Tensor, *, heads: int) -> None:
    """In place: zero K and V of references ``n >= valid_indices[b]`` (``ir_zero_invalid_refs``;
    pix2pix_turbo.py:269-273 - zeroed, not masked)."""
    _need_gpu(k, v)
    if k.dim() != 4 or k.shape != v.shape or k.stride(-1) != 1 or v.stride(-1) != 1:
        raise ValueError("k, v must be (B, N, L, H*64) views with a contiguous channel axis")
    B, N, L, _ = k.shape
    valid = torch.as_tensor(valid_indices).to(device=k.device, dtype=torch.int32).contiguous()
    if valid.numel() != B:
        raise ValueError("valid_indices must have one entry per batch element")
    rc = _lib.lib().ir_zero_invalid_refs(B, heads, N, L, valid.data_ptr(),
                                         k.data_ptr(), k.stride(0), k.stride(1), k.stride(2), HEAD_DIM,
                                         v.data_ptr(), v.stride(0), v.stride(1), v.stride(2), HEAD_DIM, _stream())
    _lib.check(rc, "ir_zero_invalid_refs")


@_on_tensor_device
def tensor2im_u8(x: torch.Tensor) -> torch.Tensor:
    """(B, 3, H, W) or (3, H, W) in [-1, 1] -> uint8 (B, H, W, 3) / (H, W, 3) on the device, with the
    exact rounding sequence of the reference's ``tensor2im(var, unnorm=True)`` (vis_utils.py:14-23)."""
    _need_gpu(x)
    squeeze = x.dim() == 3
    if squeeze:
        x = x.unsqueeze(0)
    if x.dim() != 4:
        raise ValueError("expected (B, C, H, W) or (C, H, W)")
    code = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}.get(x.dtype)
    if code is None:
        raise TypeError(f"tensor2im_u8: unsupported dtype {x.dtype}")
    B, Cc, H, W = x.shape
    out = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
    rc = _lib.lib().ir_tensor2im_u8(code, B, Cc, H, W, x.data_ptr(), x.stride(0), x.stride(1), x.stride(2),
                                    x.stride(3), out.data_ptr(), _stream())
    _lib.check(rc, "ir_tensor2im_u8")
    return out[0] if squeeze else out


LIN_AUTO, LIN_X_STATIONARY, LIN_TILED_FIRST = 0, 1, 2   # IR_LIN_* of include/instantrestore_hip.h
LIN_KERNELS = {"auto": 0, "x_stationary": 1, "256x128": 2, "128x128": 3, "128x64": 4, "256x64": 5, "64x128": 6, "128x256": 7, "256x256": 8,
               "128x128k2": 9}   # 9 (round 5): 128 x 128 tile, contraction split over two wave groups (K / 64 even)


def linear_supported(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> bool:
    """shapes / layouts ``ir_linear_fwd`` implements: K % 64 == 0 and N % 64 == 0 (LDS-tiled kernel), or K <= 320 / K = 640
    with N % 32 == 0 (X-stationary kernels)"""
    n, k = weight.shape
    if not (x.is_cuda and weight.dtype in _DT and (x.dtype == weight.dtype or x.dtype == torch.float32) and x.shape[-1] == k
            and weight.stride(1) == 1 and weight.stride(0) % 8 == 0 and x.numel() > 0
            and (bias is None or (bias.dtype == weight.dtype and bias.is_contiguous()))):
        return False
    rows = x.numel() // k
    key = (rows, n, k, bias is None)
    ok = _KERNEL_FOR.get(key)
    if ok is None:      # a pure function of the shape: asked once per shape, not once per launch
        if len(_KERNEL_FOR) > 4096:
            _KERNEL_FOR.clear()
        ok = _KERNEL_FOR[key] = _lib.lib().ir_linear_kernel_for(rows, n, k, 0 if bias is None else 1) >= 0
    return ok


_KERNEL_FOR = {}


def linear_kernel_for(rows: int, n: int, k: int, bias: bool) -> int:
    """IR_LIN_* id of the kernel the automatic choice launches for a shape (-1: unsupported)"""
    return int(_lib.lib().ir_linear_kernel_for(int(rows), int(n), int(k), 1 if bias else 0))


STATS_MAX_CHUNKS = 256   # partials per matrix the merge kernels take (ir_adain_affine_from_partials): 16 384 tokens in 64-row blocks


def linear_stats_rows(rows: int, n: int, k: int, bias: bool) -> int:
    """rows per statistics block of ``linear(..., stats=...)`` for a shape (``ir_linear_stats_rows``); 0: the kernel that
    serves the shape cannot leave statistics behind (``M`` not a multiple of its row block, ``N % 64 != 0``)"""
    return int(_lib.lib().ir_linear_stats_rows(int(rows), int(n), int(k), 1 if bias else 0))


class ColumnStats:
    """partial token statistics of a column range of a projection output, as ``ir_linear_fwd_stats`` leaves them:
    ``ws`` fp32 ``(M / rows, heads, 128)`` = mean[64] | M2[64] per (row block, head); ``rows`` tokens per block"""

    __slots__ = ("ws", "rows", "heads")

    def __init__(self, ws: torch.Tensor, rows: int, heads: int):
        self.ws, self.rows, self.heads = ws, int(rows), int(heads)

    def record_stream(self, stream) -> None:
        self.ws.record_stream(stream)


class RefStatsPartials:
    """AdaIN CONTENT statistics of the reference V's of one layer, still in the form the K/V-capture layer's q/k/v GEMM left
    them in: ``part`` = partials over ``batch * n_refs`` token sets of ``length`` rows.  The shared layer's affine kernel
    merges them directly (one launch per layer instead of a merge on the capture side plus the affine); ``finished()`` gives
    the ``(mean, std)`` pair of ``(B, N, H, 64)`` tensors that a per-identity cache stores.  ``valid``: int32 ``(B,)`` device
    tensor when references ``n >= valid[b]`` were zero-filled by the harvest (statistics (0, 0)), else ``None``."""

    __slots__ = ("part", "batch", "n_refs", "length", "valid", "producer", "_finished")

    def __init__(self, part: ColumnStats, batch: int, n_refs: int, length: int, valid: Optional[torch.Tensor] = None,
                 producer: Optional["torch.cuda.Stream"] = None):
        """``producer``: the HIP stream the capture layer's GEMM wrote the partials on (``AttnProcessor.stream``); a merge
        launched on another stream first waits for everything enqueued there (``finished()`` on the main stream
        could read the partials before the capture stream's GEMM tail had written them when no event was handed over)."""
        self.part, self.batch, self.n_refs, self.length, self.valid = part, int(batch), int(n_refs), int(length), valid
        self.producer = producer
        self._finished = None

    def __setattr__(self, name, value):
        if name == "valid":     # the zero fill changes which references count as all-zero: a cached merge is stale
            object.__setattr__(self, "_finished", None)
        object.__setattr__(self, name, value)

    def record_stream(self, stream) -> None:
        self.part.ws.record_stream(stream)
        if self.valid is not None:
            self.valid.record_stream(stream)

    def sync_to_current(self) -> None:
        """order the current stream behind the partials' producer (no-op on the producer's own stream)"""
        if self.producer is None or not self.part.ws.is_cuda:
            return
        cur = torch.cuda.current_stream(self.part.ws.device)
        if cur != self.producer:
            cur.wait_stream(self.producer)
            self.record_stream(cur)

    def finished(self):
        """``(mean, std)`` of every reference V, fp32 ``(B, N, H, 64)``: merged once, then cached on the object"""
        if self._finished is not None:
            return self._finished
        self.sync_to_current()
        mean, std = token_stats_from_partials(self.part, self.batch * self.n_refs, self.length)
        mean, std = mean.reshape(self.batch, self.n_refs, *mean.shape[-2:]), std.reshape(self.batch, self.n_refs, *std.shape[-2:])
        if self.valid is not None:
            keep = (torch.arange(self.n_refs, device=mean.device)[None, :] < self.valid.reshape(-1, 1)).to(torch.float32)[:, :, None, None]
            mean, std = mean * keep, std * keep
        self._finished = (mean.contiguous(), std.contiguous())
        return self._finished


@_on_tensor_device
def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, scale_cols: int = 0,
           col_scale: float = 1.0, kernel: int = 0, stats: Optional[Tuple[int, int]] = None):
    """``F.linear(x, weight, bias)`` for ``x (..., K)``, 16-bit ``weight (N, K)`` (``ir_linear_fwd_ex``): fp32 accumulation,
    one rounding.  Raises for unsupported shapes.  ``scale_cols`` / ``col_scale``: the first ``scale_cols`` output columns
    are multiplied by ``col_scale`` in fp32 before that rounding.  ``x`` may be fp32: it is rounded to the weight's dtype
    while loaded (the autocast cast, fused).  ``kernel``: ``LIN_KERNELS`` (0 = the library's own choice).

    ``stats=(col0, cols)``: also return the partial token statistics of output columns ``[col0, col0 + cols)`` - whole
    heads - computed by the GEMM's own workgroups from the block they have just stored (``ir_linear_fwd_stats``; the AdaIN
    statistics without a pass over V): the result is ``(y, ColumnStats)``.  Check :func:`linear_stats_rows` first."""
    _need_gpu(x, weight, bias)
    _forward_only(x, weight, bias)
    n, k = weight.shape
    x2 = x.reshape(-1, k)
    if x2.stride(1) != 1 or x2.stride(0) % 8 != 0:
        x2 = x2.contiguous()
    y = torch.empty((x2.shape[0], n), dtype=weight.dtype, device=x.device)
    if stats is not None:
        if kernel:
            raise ValueError("stats: the kernel choice is the library's (kernel=0)")
        col0, cols = int(stats[0]), int(stats[1])
        rows = linear_stats_rows(x2.shape[0], n, k, bias is not None)
        if rows <= 0 or cols <= 0 or cols % HEAD_DIM or col0 % HEAD_DIM:
            raise ValueError(f"linear(stats=...): shape ({x2.shape[0]}, {n}, {k}) / columns ({col0}, {cols}) cannot carry the statistics tail")
        ws = torch.empty((x2.shape[0] // rows, cols // HEAD_DIM, 128), dtype=torch.float32, device=x.device)
        rc = _lib.lib().ir_linear_fwd_stats(_dtype_code(weight), 1 if x.dtype == torch.float32 else 0, x2.shape[0], n, k, x2.data_ptr(),
                                            x2.stride(0), weight.data_ptr(), weight.stride(0),
                                            None if bias is None else bias.data_ptr(), y.data_ptr(), n, int(scale_cols),
                                            float(col_scale), col0, cols, ws.data_ptr(), ws.numel() * 4, _stream())
        _lib.check(rc, "ir_linear_fwd_stats")
        return y.view(*x.shape[:-1], n), ColumnStats(ws, rows, cols // HEAD_DIM)
    rc = _lib.lib().ir_linear_fwd_ex(_dtype_code(weight), 1 if x.dtype == torch.float32 else 0, x2.shape[0], n, k, x2.data_ptr(), x2.stride(0), weight.data_ptr(),
                                     weight.stride(0), None if bias is None else bias.data_ptr(), y.data_ptr(), n,
                                     int(scale_cols), float(col_scale), int(kernel), _stream())
    _lib.check(rc, "ir_linear_fwd_ex")
    return y.view(*x.shape[:-1], n)


@_on_tensor_device
def adain_affine_from_partials(style: ColumnStats, batch: int, len_self: int, n_refs: int, len_ref: int, *,
                               content: Optional[ColumnStats] = None, content_mean: Optional[torch.Tensor] = None,
                               content_std: Optional[torch.Tensor] = None, valid: Optional[torch.Tensor] = None,
                               eps: float = ADAIN_EPS):
    """The AdaIN affine ``(a, b)`` of :func:`adain_stats` from the partials the projections left behind
    (``ir_adain_affine_from_partials``): ``style`` from the shared layer's q/k/v GEMM over ``batch`` sets of ``len_self``
    tokens; the content statistics either as the K/V-capture layer's partials (``content``, ``batch * n_refs`` sets of
    ``len_ref`` tokens) or finished (``content_mean`` / ``content_std``, fp32 ``(B, N, H, 64)``).  ``valid``: int32 ``(B)``
    on the device, references ``n >= valid[b]`` count as zero-filled."""
    H = style.heads
    dev = style.ws.device
    if (content is None) == (content_mean is None or content_std is None):
        raise ValueError("content statistics: either `content` partials or content_mean + content_std")
    if content is not None and content.heads != H:
        raise ValueError("style / content head counts differ")
    for t in (content_mean, content_std):
        if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (batch, n_refs, H, HEAD_DIM)):
            raise ValueError(f"content statistics must be contiguous fp32 ({batch}, {n_refs}, {H}, 64)")
    if style.ws.shape[0] * style.rows != batch * len_self or (content is not None and content.ws.shape[0] * content.rows != batch * n_refs * len_ref):
        raise ValueError("partials do not cover batch x len rows")
    if valid is not None and (valid.dtype != torch.int32 or valid.device != dev or valid.numel() != batch or not valid.is_contiguous()):
        raise ValueError(f"valid must be a contiguous int32 ({batch},) tensor on {dev}")
    a = torch.empty((batch, n_refs, H, HEAD_DIM), dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    rc = _lib.lib().ir_adain_affine_from_partials(
        batch, H, n_refs, len_self, len_ref, style.ws.data_ptr(), style.rows,
        None if content is None else content.ws.data_ptr(), 0 if content is None else content.rows,
        None if content_mean is None else content_mean.data_ptr(), None if content_std is None else content_std.data_ptr(),
        None if valid is None else valid.data_ptr(), float(eps), a.data_ptr(), b.data_ptr(), _stream())
    _lib.check(rc, "ir_adain_affine_from_partials")
    return a, b


@_on_tensor_device
def token_stats_from_partials(part: ColumnStats, n_sets: int, length: int):
    """mean and unbiased std over the tokens of ``n_sets`` matrices of ``length`` rows from their partials
    (``ir_token_stats_from_partials``): two fp32 ``(n_sets, H, 64)`` tensors, what :func:`token_stats` returns per matrix"""
    if part.ws.shape[0] * part.rows != n_sets * length:
        raise ValueError("partials do not cover n_sets x length rows")
    mean = torch.empty((n_sets, part.heads, HEAD_DIM), dtype=torch.float32, device=part.ws.device)
    std = torch.empty_like(mean)
    rc = _lib.lib().ir_token_stats_from_partials(n_sets, part.heads, length, part.ws.data_ptr(), part.rows, mean.data_ptr(),
                                                 std.data_ptr(), _stream())
    _lib.check(rc, "ir_token_stats_from_partials")
    return mean, std


_ANY_DT = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}


@_on_tensor_device
def freeu_fourier_filter(x: torch.Tensor, threshold: int, scale: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FreeU's skip-feature filter on (B, C, H, W) (``ir_freeu_fourier_filter``; replaces
    ``fourier_filter(x.float(), threshold, scale).to(x.dtype)``, block.py:3514,3518).  Result has
    ``x``'s dtype; ``out=x`` filters in place."""
    _need_gpu(x, out)
    if x.dim() != 4:
        raise ValueError("expected (B, C, H, W)")
    code = _ANY_DT.get(x.dtype)
    if code is None:
        raise TypeError(f"freeu_fourier_filter: unsupported dtype {x.dtype}")
    B, Cc, H, W = x.shape
    if not (x.stride(3) == 1 and x.stride(2) == W and (B == 1 or x.stride(0) == Cc * x.stride(1))):
        x = x.contiguous()
    if out is None:
        out = torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device)
    elif out.shape != x.shape or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError("out must be a contiguous tensor of x's shape and dtype")
    rc = _lib.lib().ir_freeu_fourier_filter(code, B * Cc, H, W, x.data_ptr(), x.stride(1), out.data_ptr(), H * W,
                                            int(threshold), float(scale), _stream())
    _lib.check(rc, "ir_freeu_fourier_filter")
    return out


def lanczos_coeffs(in_size: int, out_size: int):
    """Pillow's LANCZOS tap tables for one axis, computed on the host by the library
    (``ir_lanczos_coeffs``): ``(bounds (out, 2) int32, kk (out, ksize) int32)`` CPU tensors."""
    L = _lib.lib()
    ksize = L.ir_lanczos_ksize(int(in_size), int(out_size))
    if ksize <= 0:
        _lib.check(ksize, "ir_lanczos_ksize")
    bounds = torch.empty((out_size, 2), dtype=torch.int32)
    kk = torch.empty((out_size, ksize), dtype=torch.int32)
    _lib.check(L.ir_lanczos_coeffs(int(in_size), int(out_size), bounds.data_ptr(), kk.data_ptr()), "ir_lanczos_coeffs")
    return bounds, kk


def preprocess_lanczos(descs, size: int, dtype: torch.dtype, device: torch.device) -> torch.Tensor:
    """Launch ``ir_preprocess_lanczos_u8`` over a ctypes array of ``_lib.ImageDesc`` (built by
    ``instantrestore_amd.preprocess``); returns ``(n, 3, size, size)`` in ``dtype``."""
    code = _ANY_DT.get(dtype)
    if code is None:
        raise TypeError(f"preprocess: unsupported output dtype {dtype}")
    n = len(descs)
    with torch.cuda.device(device):
        out = torch.empty((n, 3, size, size), dtype=dtype, device=device)
        rc = _lib.lib().ir_preprocess_lanczos_u8(descs, n, int(size), code, out.data_ptr(), _stream())
    _lib.check(rc, "ir_preprocess_lanczos_u8")
    return out


_TUNING = int(os.environ.get("IR_ATTN_VARIANT", "0") or 0)


def tuning_supports_prescaled_q() -> bool:
    """``IR_FLAG_Q_PRESCALED`` is implemented by the default dispatch and by the kernels it picks from (tuning 0, 11, 13, 16, 18;
    the development build's ablations 20-28 take the flag explicitly); under any other A/B ``tuning`` the processors keep the plain q (the C ABI rejects the combination)"""
    return _TUNING in (0, 11, 13, 16, 18)


def set_attn_variant(variant: int) -> int:
    """tuning hook for benchmarks / A-B tests: the value goes into the per-call ``tuning`` field of the C ABI's
    argument block (``IR_TUNE_*`` in include/instantrestore_hip.h; 0 = default dispatch, 16 = 128-row kernel (one wave per SIMD; pre-scaled Q only), 13 / 12 = 64-row kernel in 8- / 4-wave workgroups, 10 / 14 = pipelined 32-row kernel, 11 = its
    pre-scaled-Q form).  ``IR_ATTN_VARIANT=<n>`` in the environment sets the initial value.  Returns the previous one.
    The C library itself holds no such state."""
    global _TUNING
    prev, _TUNING = _TUNING, int(variant)
    return prev
