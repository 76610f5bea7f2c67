"""Reference K/V harvest: the path-relevant part of
``Pix2Pix_Turbo.get_conditioning_keys_values`` (face_replace/models/pix2pix_turbo.py:260-275).

After the frozen reference UNet has run over the ``B*N`` reference latents, each of its nine
decoder self-attention processors (:class:`AttnProcessor`, selected by EXACT type like the
reference does at :260) holds ``keys`` / ``values`` of shape ``(B*N, L, C)``.  They are
re-viewed as ``(B, N, L, C)`` - a view, never a copy - the references ``n >= valid_indices[b]``
are zero-filled in place (zeroed, NOT masked: they keep their exp(0) softmax weight, :269-273)
by one HIP launch per layer instead of the reference's Python ``layers x samples`` loop, and the
processors are reset (:275).  The two lists are what the main UNet receives as
``cross_attention_kwargs={'ref_keys': ..., 'ref_values': ...}``.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from . import attn_processors as _ap
from . import ops as _ops


def harvest_reference_kv(original_unet, n_refs: int, valid_indices: Sequence[int],
                         reset: bool = True, with_events: bool = False, with_stats: bool = False, with_valid: bool = False):
    """``(keys, values)`` - or ``(keys, values, events)`` with ``with_events``: one HIP event per layer,
    recorded by the capturing processor right after its K/V projections when ``record_events`` was set on
    it (see :func:`enable_stream_overlap`); hand them to the main UNet as
    ``cross_attention_kwargs={'ref_keys': ..., 'ref_values': ..., 'ref_events': ...}`` when the two UNets
    run on different streams.

    ``with_stats`` appends ``stats``: per layer either ``ops.RefStatsPartials`` (round 4: the statistics as the capture
    layer's q/k/v GEMM left them; ``.finished()`` gives the pair below) or ``(mean, std)`` of every reference V over its tokens,
    fp32 ``(B, N, H, 64)`` - the CONTENT statistics of AdaIN (attn_processors.py:9-10), stashed by the capturing processors
    (:func:`enable_ref_stats`) or computed here when they were not; pass it on as ``'ref_stats'`` and the shared layers
    read only their own V (``ir_adain_stats_cached``).  References zero-filled below get the statistics of an all-zero
    V, (0, 0): exactly what the uncached path computes from the zeroed tensor (the ``b == mean(V_self)`` quirk).

    ``with_valid`` (round 5) appends ``valid``: the int32 ``(B,)`` device tensor of valid counts when some reference was
    zero-filled here, else ``None`` - hand it on as ``'ref_valid'`` and the shared layers close the zeroed segments in closed
    form instead of walking them (ABI v8 ``valid_refs``; same output)."""
    procs = [p for p in original_unet.attn_processors.values() if type(p) in [_ap.AttnProcessor]]
    if not procs:
        raise RuntimeError("no AttnProcessor on this UNet: call register_attention_processor_kv_unet first")
    keys, values, events, streams, stats = [], [], [], [], []
    for p in procs:
        if p.keys is None or p.values is None:
            raise RuntimeError("reference UNet has not been run since the last reset()")
        events.append(p.ready)
        streams.append(getattr(p, "stream", None))
        k = p.keys.reshape(-1, n_refs, p.keys.shape[1], p.keys.shape[2])
        v = p.values.reshape(-1, n_refs, p.values.shape[1], p.values.shape[2])
        keys.append(k)
        values.append(v)
        if with_stats and getattr(p, "v_part", None) is not None:
            # round 4: the capture layer's q/k/v GEMM left the partial statistics of its V third behind; they travel as they
            # are (the shared layer's affine kernel merges them), with the valid counts when references get zero-filled below
            stats.append(_ops.RefStatsPartials(p.v_part, v.shape[0], n_refs, v.shape[2], producer=streams[-1]))
        elif with_stats:
            m, sd = getattr(p, "v_mean", None), getattr(p, "v_std", None)
            if m is None and v.is_cuda:      # not stashed at capture time: one pass over V now, behind its producer
                cur = torch.cuda.current_stream(v.device)
                if streams[-1] is not None and streams[-1] != cur:
                    cur.wait_stream(streams[-1])
                    v.record_stream(cur)
                heads = v.shape[-1] // _ops.HEAD_DIM
                m, sd = _ops.token_stats(v, heads=heads)
            stats.append(None if m is None else (m.reshape(-1, n_refs, *m.shape[-2:]), sd.reshape(-1, n_refs, *sd.shape[-2:])))
    valid = torch.as_tensor(valid_indices)
    valid_out = None
    if bool((valid < n_refs).any()):
        if keys[0].is_cuda:
            # The zero fill runs on the CURRENT stream and rewrites the stashes in place.  The reference does it after
            # the whole reference forward (pix2pix_turbo.py:255-273); when that forward was enqueued on another stream
            # (two-stream pipelining) the per-layer `ready` events are NOT enough - they are recorded before the
            # layer's own attention reads K/V - so this stream first waits for everything the capture stream(s) have
            # been given, i.e. the end of the reference forward.
            cur = torch.cuda.current_stream(keys[0].device)
            for st in {s for s in streams if s is not None and s != cur}:
                cur.wait_stream(st)
            for k, v in zip(keys, values):   # allocated on the capture stream, written here
                k.record_stream(cur)
                v.record_stream(cur)
        for k, v in zip(keys, values):
            _ops.zero_invalid_refs(k, v, valid, heads=k.shape[-1] // _ops.HEAD_DIM)
        valid_out = valid.to(device=keys[0].device, dtype=torch.int32).contiguous()
        if with_stats and keys[0].is_cuda:
            # the zero fill invalidates the cached statistics of the zeroed references: an all-zero V has mean 0, std 0
            keep = (torch.arange(n_refs)[None, :] < valid.reshape(-1, 1)).to(device=keys[0].device, dtype=torch.float32)[:, :, None, None]
            valid_dev = valid_out
            for st in stats:
                if hasattr(st, "finished"):
                    st.valid = valid_dev          # the affine kernel (and finished()) count these references as all-zero
                elif st is not None:
                    for t in st:
                        t.record_stream(torch.cuda.current_stream(t.device))
                        t.mul_(keep)
    if reset:
        for p in procs:
            p.reset()
    if with_events:
        if bool((valid < n_refs).any()):
            # the zero fill above ran on the CURRENT stream after the captures: make it the thing to wait for
            ev = torch.cuda.Event()
            ev.record()
            events = [ev] * len(keys)
        res = (keys, values, events, stats) if with_stats else (keys, values, events)
    else:
        res = (keys, values, stats) if with_stats else (keys, values)
    return res + (valid_out,) if with_valid else res


def finished_stats(stats):
    """the harvested AdaIN content statistics as per-layer ``(mean, std)`` pairs of fp32 ``(B, N, H, 64)`` tensors - what a
    per-identity cache stores and slices - whichever form the harvest handed them over in (``ops.RefStatsPartials`` when the
    capture GEMM produced them, pairs already otherwise)"""
    return [st.finished() if hasattr(st, "finished") else st for st in stats]


def enable_ref_stats(original_unet, enabled: bool = True) -> None:
    """make every K/V-capturing processor also stash the AdaIN content statistics of its V (``AttnProcessor.v_mean`` /
    ``v_std``): they are computed once per reference, on the stream the reference UNet runs on, and
    :func:`harvest_reference_kv` ``(with_stats=True)`` hands them to the main UNet"""
    for p in original_unet.attn_processors.values():
        if type(p) in [_ap.AttnProcessor]:
            p.capture_stats = bool(enabled)


def enable_stream_overlap(original_unet, enabled: bool = True) -> None:
    """make every K/V-capturing processor of the reference UNet record a HIP event when its K/V are
    stashed, so the main UNet (on another stream) can start each shared layer as soon as ITS reference
    layer is done instead of after the whole reference forward"""
    for p in original_unet.attn_processors.values():
        if type(p) in [_ap.AttnProcessor]:
            p.record_events = bool(enabled)


def get_conditioning_keys_values(original_unet, model_input: torch.Tensor, timestep, encoder_hidden_states,
                                 n_refs: int, valid_indices: Sequence[int], early_exit: bool = False,
                                 with_stats: bool = False, with_valid: bool = False):
    """Run the frozen reference UNet on the (already encoded and noised) reference latents
    ``(B*N, 4, S, S)`` and harvest.  VAE encode/decode, the scheduler and the caption encoder
    around it (pix2pix_turbo.py:244-257, 277-278) are stock PyTorch and out of scope.

    ``early_exit`` (off by default): the reference UNet's own output is
    thrown away by the inference caller (``inference/test.py:100`` keeps only the K/V lists), so its
    forward can stop at the last K/V-capturing layer - after ``to_k`` / ``to_v`` of that layer, before its
    attention, its out projection and everything downstream.  The harvested lists are identical."""
    procs = [p for p in original_unet.attn_processors.values() if type(p) in [_ap.AttnProcessor]]
    if not procs:
        raise RuntimeError("no AttnProcessor on this UNet: call register_attention_processor_kv_unet first")
    # the capture layers compute the content statistics for THIS call iff it asks for them; whatever the caller had set
    # with enable_ref_stats comes back afterwards (a with_stats=True call must not leave every later capture paying for them)
    saved_stats = [p.capture_stats for p in procs]
    for p in procs:
        p.capture_stats = bool(with_stats) or p.capture_stats
    try:
        if not early_exit:
            original_unet(model_input, timestep, encoder_hidden_states=encoder_hidden_states)
            return harvest_reference_kv(original_unet, n_refs, valid_indices, with_stats=with_stats, with_valid=with_valid)
        for p in procs:                          # whichever capturing layer runs last stops the forward
            p.reset()
            p.stop_after_capture = procs
        try:
            original_unet(model_input, timestep, encoder_hidden_states=encoder_hidden_states)
        except _ap.ReferenceCaptureComplete:
            pass
        else:
            raise RuntimeError("early exit armed but the capturing layers did not all run")
        finally:
            for p in procs:
                p.stop_after_capture = None
        return harvest_reference_kv(original_unet, n_refs, valid_indices, with_stats=with_stats, with_valid=with_valid)
    finally:
        for p, flag in zip(procs, saved_stats):
            p.capture_stats = flag


# ------------------------------------------------------------------------------------------------------------------------
# round 6: the step as ONE hipGraph (inference/test.py:79-111 runs eagerly with B = 1: ~80 small launches whose issue time
# on the host exceeds their run time on the device - NOTES 11.11)
# ------------------------------------------------------------------------------------------------------------------------
class CapturedStep:
    """``fn()`` - both UNets' attention path: reference capture -> harvest -> main UNet - recorded once into ONE hipGraph.

    ``fn`` takes no arguments: it reads STATIC input tensors the caller owns (refill them with ``copy_`` before a replay) and
    returns tensors (or nested lists / tuples of tensors), which become the graph's static results: ``result`` after every
    ``replay()``.  Streams forked inside ``fn`` (the two-stream schedule of :func:`enable_stream_overlap`) are captured with
    it as long as they are joined back through events before ``fn`` returns, which the processors' ``ref_events`` do.

    Capture rules the helper takes care of: ``fn`` runs ``warmup`` times eagerly on the capture stream first (lazy
    initialisation - scratch buffers, folded weights, kernel attributes - must not happen inside the capture; the scratch
    buffers handed out during the capture are pinned for the life of the thread, ``ops._workspace``), and the capture stream
    is not the default stream.  Shapes are frozen: one ``CapturedStep`` per (B, N, px) - :class:`StepGraphs` keeps them."""

    def __init__(self, fn, warmup: int = 1, stream: "torch.cuda.Stream | None" = None):
        if not torch.cuda.is_available():
            raise RuntimeError("capture_step needs the GPU (the product has no CPU path)")
        self.graph = torch.cuda.CUDAGraph()
        self.stream = stream if stream is not None else torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream), torch.no_grad():
            for _ in range(max(1, int(warmup))):
                fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.result = fn()
        torch.cuda.current_stream().wait_stream(self.stream)
        self.replays = 0

    def replay(self):
        """launch the recorded step on the current stream; returns the static results (valid once the stream reaches them)"""
        self.graph.replay()
        self.replays += 1
        return self.result


def capture_step(fn, warmup: int = 1, stream=None) -> CapturedStep:
    """record ``fn()`` into one hipGraph (see :class:`CapturedStep`)"""
    return CapturedStep(fn, warmup=warmup, stream=stream)


class StepGraphs:
    """one captured step per shape key, e.g. ``(B, N, px, dtype)``: ``graphs.run(key, make_fn)`` captures on first use -
    ``make_fn()`` returns the zero-argument step closed over that shape's static inputs - and replays afterwards"""

    def __init__(self, max_entries: int = 8):
        self._g = {}
        self.max_entries = max_entries

    def run(self, key, make_fn, warmup: int = 1):
        g = self._g.get(key)
        if g is None:
            if len(self._g) >= self.max_entries:          # oldest capture first (dict order); its graph and static tensors go
                self._g.pop(next(iter(self._g)))
            g = self._g[key] = capture_step(make_fn(), warmup=warmup)
        return g.replay()

    def __len__(self):
        return len(self._g)
