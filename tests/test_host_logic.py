"""Host-side logic on CPU (no GPU, no HIP compute): the drop-in boundary of SURVEY.md section 8b.
The compute entry points are replaced by the oracle-backed ``tests/oracle_ops.py`` - the product
itself has no CPU path (see test_product_has_no_cpu_fallback)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import oracle_ops
from conftest import GOLDEN_MANIFEST
from oracle import shared_attn_oracle as O


@pytest.fixture()
def shim(monkeypatch):
    import instantrestore_amd.attn_processors as ap
    import instantrestore_amd.kv_harvest as kh
    monkeypatch.setattr(ap, "_ops", oracle_ops)
    monkeypatch.setattr(kh, "_ops", oracle_ops)
    oracle_ops.CALLS.clear()
    return oracle_ops


CFG = SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False)


def test_module_path_and_six_names():
    import face_replace.models.attn_processors as m
    import instantrestore_amd.attn_processors as impl
    for name in ("adain", "AttnProcessor", "FaceIDAttnProcessor", "SharedAttnProcessor",
                 "register_attention_processor", "register_attention_processor_kv_unet"):
        assert getattr(m, name) is getattr(impl, name)
    # constructor signatures of the reference (attn_processors.py:25,102,186)
    p = m.SharedAttnProcessor()
    assert (p.self_attn_idx, p.save_self_attentions, p.use_adain, p.train_input) == (None, False, False, True)
    f = m.FaceIDAttnProcessor(hidden_size=128, cross_attention_dim=96)
    assert f.face_projection.in_features == 512 and f.to_k_face_embed.out_features == 128
    k = m.AttnProcessor()
    assert k.keys is None and k.values is None and k.is_self_attn is None
    # no parameters / buffers on the processors that ship in checkpoints (strict load, test.py:47-50)
    assert len(p.state_dict()) == 0 and len(k.state_dict()) == 0


def test_registration_assigns_nine_indices_in_order():
    from face_replace.models.attn_processors import (AttnProcessor, FaceIDAttnProcessor, SharedAttnProcessor,
                                                     register_attention_processor,
                                                     register_attention_processor_kv_unet)
    from instantrestore_amd.unet_host import AttnTopologyUNet
    unet = AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                            cross_attention_dim=64)
    names = list(unet.attn_processors.keys())
    assert len(names) == 32 and names[0].startswith("down_blocks.0") and names[-1].startswith("mid_block")
    with pytest.raises(ValueError):
        unet.set_attn_processor({names[0]: SharedAttnProcessor()})
    register_attention_processor(unet, CFG, save_self_attentions=True)
    procs = unet.attn_processors
    idx = [(n, p.self_attn_idx) for n, p in procs.items() if p.self_attn_idx is not None]
    assert [i for _, i in idx] == list(range(9))
    assert [n.split(".attentions")[0] for n, _ in idx] == ["up_blocks.1"] * 3 + ["up_blocks.2"] * 3 + ["up_blocks.3"] * 3
    assert all(type(p) == SharedAttnProcessor for p in procs.values())          # exact-type checks of the callers
    assert all(p.use_adain and p.train_input for p in procs.values())
    assert all(p.save_self_attentions for n, p in procs.items() if n.endswith("attn1.processor"))
    assert not any(p.save_self_attentions for n, p in procs.items() if n.endswith("attn2.processor"))
    assert len(unet.state_dict()) == len(AttnTopologyUNet(block_out_channels=(64, 128, 128, 128),
                                                          attention_head_dim=(1, 2, 2, 2),
                                                          cross_attention_dim=64).state_dict())
    # face-id variant
    face_cfg = SimpleNamespace(use_adain=False, train_input=False, condition_on_face_embeds=True)
    register_attention_processor(unet, face_cfg)
    assert all(type(p) == FaceIDAttnProcessor for n, p in unet.attn_processors.items() if n.endswith("attn2.processor"))
    # kv unet: only decoder self-attentions are replaced, the rest keep their object identity
    register_attention_processor(unet, CFG)
    before = unet.attn_processors
    register_attention_processor_kv_unet(unet)
    after = unet.attn_processors
    kv = [n for n, p in after.items() if type(p) in [AttnProcessor]]
    assert len(kv) == 9 and all(n.startswith("up_blocks") and "attn1" in n for n in kv)
    assert all(after[n] is before[n] for n in after if n not in kv)


def test_two_unet_call_pattern_end_to_end(shim):
    """pix2pix_turbo.py:255-275 + :322-326 on a small host: harvest views, zero fill shows through
    to the stash, reset, kwargs reach every attention, one shared layer checked against the oracle."""
    from face_replace.models.attn_processors import (AttnProcessor, SharedAttnProcessor, register_attention_processor,
                                                     register_attention_processor_kv_unet)
    from instantrestore_amd.kv_harvest import harvest_reference_kv
    from instantrestore_amd.unet_host import AttnTopologyUNet
    mk = lambda seed: AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                                       cross_attention_dim=32, seed=seed)
    kv_unet, unet = mk(1), mk(2)
    kv_unet.set_attn_processor({n: SharedAttnProcessor(self_attn_idx=None) for n in kv_unet.attn_processors})
    register_attention_processor_kv_unet(kv_unet)
    cfg = SimpleNamespace(use_adain=True, train_input=False, condition_on_face_embeds=False)
    register_attention_processor(unet, cfg, save_self_attentions=True)
    B, N, S = 2, 3, 8
    text = torch.randn(1, 5, 32)
    with torch.no_grad():
        kv_unet(torch.randn(B * N, 4, S, S), None, encoder_hidden_states=text.repeat(B * N, 1, 1))
        stash = [p for p in kv_unet.attn_processors.values() if type(p) in [AttnProcessor]]
        assert all(p.keys.shape[0] == B * N and p.is_self_attn for p in stash)
        raw_k = [p.keys for p in stash]
        keys, vals = harvest_reference_kv(kv_unet, N, [3, 1], reset=False)
        assert len(keys) == 9 and [tuple(k.shape[:2]) for k in keys] == [(B, N)] * 9
        assert [k.shape[2] for k in keys] == [4] * 3 + [16] * 3 + [64] * 3        # 2x2, 4x4, 8x8 tokens
        for k, rk in zip(keys, raw_k):
            assert k.data_ptr() == rk.data_ptr()                                  # view, not a copy
            assert torch.all(k[1, 1:] == 0) and torch.any(k[1, 0] != 0) and torch.any(k[0, 2] != 0)
            assert torch.all(rk.reshape(B, N, *rk.shape[1:])[1, 1:] == 0)         # stash itself zeroed
        for p in stash:
            p.reset()
        assert all(p.keys is None for p in stash)
        # grab the input of the first 8x8 decoder layer to check it against the oracle
        target = unet.decoder_self_attentions()[6]
        seen = {}
        hook = target.register_forward_pre_hook(lambda m, args, kwargs: seen.update(h=args[0], kw=kwargs), with_kwargs=True)
        y = unet(torch.randn(B, 4, S, S), None, encoder_hidden_states=text.repeat(B, 1, 1),
                 cross_attention_kwargs={"ref_keys": keys, "ref_values": vals}).sample
        hook.remove()
    assert y.shape == (B, 4, S, S) and torch.isfinite(y).all()
    assert set(seen["kw"]) == {"ref_keys", "ref_values"} and seen["kw"]["ref_keys"] is keys
    shared = [p for p in unet.attn_processors.values() if type(p) == SharedAttnProcessor and p.self_attn_idx is not None]
    assert len(shared) == 9
    for p, k in zip(shared, keys):   # (B, H, L, Lkv) with Lkv = N*L because train_input is False
        assert tuple(p.attention_probs.shape) == (B, k.shape[-1] // 64, k.shape[2], N * k.shape[2])
    names = [c[0] for c in shim.CALLS]
    assert names.count("adain_stats") == 9 and names.count("attn_probs") >= 9
    # oracle on the hooked layer
    h = seen["h"]
    w = lambda lin: lin.weight.detach().numpy()
    ref = O.shared_attn_processor_np(h.numpy(), w(target.to_q), w(target.to_k), w(target.to_v), w(target.to_out[0]),
                                     target.to_out[0].bias.detach().numpy(), keys[6].numpy(), vals[6].numpy(),
                                     target.heads, use_adain=True, train_input=False)
    with torch.no_grad():
        got = target(h, ref_keys=keys, ref_values=vals)
    np.testing.assert_allclose(got.numpy(), ref, atol=2e-5, rtol=1e-4)


def test_valid_counts_travel_from_the_harvest_to_the_kernel_call(shim):
    """round 5 (ABI v8): ``harvest_reference_kv(with_valid=True)`` hands out the int32 valid counts when it zero-filled something
    (None otherwise); as ``cross_attention_kwargs['ref_valid']`` they reach every SHARED layer's kernel call and no other; the
    opt-in ``save_attention_mass`` leaves the block sums of the probabilities"""
    from face_replace.models.attn_processors import (SharedAttnProcessor, register_attention_processor,
                                                     register_attention_processor_kv_unet)
    from instantrestore_amd.kv_harvest import harvest_reference_kv
    from instantrestore_amd.unet_host import AttnTopologyUNet
    mk = lambda seed: AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                                       cross_attention_dim=32, seed=seed)
    kv_unet, unet = mk(3), mk(4)
    kv_unet.set_attn_processor({n: SharedAttnProcessor(self_attn_idx=None) for n in kv_unet.attn_processors})
    register_attention_processor_kv_unet(kv_unet)
    register_attention_processor(unet, SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False))
    B, N, S = 2, 3, 8
    text = torch.randn(1, 5, 32)
    with torch.no_grad():
        kv_unet(torch.randn(B * N, 4, S, S), None, encoder_hidden_states=text.repeat(B * N, 1, 1))
        keys, vals, valid = harvest_reference_kv(kv_unet, N, [3, 3], reset=False, with_valid=True)
        assert valid is None                                   # nothing zero-filled: nothing to promise
        keys, vals, valid = harvest_reference_kv(kv_unet, N, [2, 1], with_valid=True)
        assert valid.dtype == torch.int32 and valid.tolist() == [2, 1]
        shared = [p for p in unet.attn_processors.values() if type(p) == SharedAttnProcessor and p.self_attn_idx is not None]
        shared[0].save_attention_mass = True
        del shim.CALLS[:]
        y = unet(torch.randn(B, 4, S, S), None, encoder_hidden_states=text.repeat(B, 1, 1),
                 cross_attention_kwargs={"ref_keys": keys, "ref_values": vals, "ref_valid": valid}).sample
    assert torch.isfinite(y).all()
    calls = [c[1] for c in shim.CALLS if c[0] == "shared_attention"]
    # the shared layers are told the counts - except the three whose reference axis has 2 x 2 = 4 tokens on this 8 x 8 latent: below
    # FOLD_MIN_REF_TOKENS the processor applies AdaIN to V (ir_adain_apply) instead of folding it, the zero-filled references' V is
    # then the style mean, not zero, and the promise no longer holds for that call
    assert sum(1 for c in calls if c["valid_refs"] == [2, 1]) == 6 and sum(1 for c in calls if c["valid_refs"] is None and c["n_refs"] == N) == 3
    assert sum(1 for c in shim.CALLS if c[0] == "adain_apply") == 3
    assert all(c["valid_refs"] is None for c in calls if c["n_refs"] == 0)    # plain / cross attention: never
    m = shared[0].attention_mass
    assert m.shape[0] == B and m.shape[-1] == N + 1 and float((m.sum(-1) - 1).abs().max()) < 1e-5
    # a zero-filled reference keeps its exp(0) weight: its mass is NOT zero (zeroed, not masked; pix2pix_turbo.py:269-273)
    assert float(m[1, ..., 2:].min()) > 0


def test_face_embed_processors_accept_the_harvests_valid_counts(shim):
    """round 6: with ``condition_on_face_embeds`` every cross-attention is a ``FaceIDAttnProcessor`` and receives the SAME
    ``cross_attention_kwargs`` as the shared layers (unet.py forwards one dict to every processor) - ``ref_valid`` included"""
    from face_replace.models.attn_processors import (FaceIDAttnProcessor, SharedAttnProcessor, register_attention_processor,
                                                     register_attention_processor_kv_unet)
    from instantrestore_amd.kv_harvest import harvest_reference_kv
    from instantrestore_amd.unet_host import AttnTopologyUNet
    mk = lambda seed: AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                                       cross_attention_dim=32, seed=seed)
    kv_unet, unet = mk(5), mk(6)
    kv_unet.set_attn_processor({n: SharedAttnProcessor(self_attn_idx=None) for n in kv_unet.attn_processors})
    register_attention_processor_kv_unet(kv_unet)
    register_attention_processor(unet, SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=True))
    face = [p for p in unet.attn_processors.values() if type(p) == FaceIDAttnProcessor]
    assert len(face) == 16
    B, N, S = 2, 3, 8
    with torch.no_grad():
        kv_unet(torch.randn(B * N, 4, S, S), None, encoder_hidden_states=torch.randn(B * N, 5, 32))
        keys, vals, valid = harvest_reference_kv(kv_unet, N, [2, 1], with_valid=True)
        del shim.CALLS[:]
        # the face embedding stands where the text states stand (model.py hands (B, 1, 512) embeddings to the cross-attentions)
        y = unet(torch.randn(B, 4, S, S), None, encoder_hidden_states=torch.randn(B, 1, 512),
                 cross_attention_kwargs={"ref_keys": keys, "ref_values": vals, "ref_valid": valid}).sample
    assert torch.isfinite(y).all()
    assert all(p.is_self_attn is False for p in face)
    calls = [c[1] for c in shim.CALLS if c[0] == "shared_attention"]
    assert sum(1 for c in calls if c["valid_refs"] == [2, 1]) == 6     # the shared layers still get the counts


SHARED = [m for m in GOLDEN_MANIFEST if m["kind"] == "shared"]


@pytest.mark.parametrize("m", SHARED, ids=[m["id"] for m in SHARED])
def test_processor_host_logic_matches_reference_outputs(shim, golden, m):
    """our processor classes (projections, branch selection, epilogue) + oracle compute must give
    the reference's golden outputs: pins the HOST side of the drop-in."""
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    C = m["H"] * 64
    attn = Attention(query_dim=C, cross_attention_dim=m.get("cross_dim"), heads=m["H"], dim_head=64)
    with torch.no_grad():
        for lin, name in ((attn.to_q, "wq"), (attn.to_k, "wk"), (attn.to_v, "wv"), (attn.to_out[0], "wo")):
            lin.weight.copy_(torch.from_numpy(golden.arr(m, name)))
        attn.to_out[0].bias.copy_(torch.from_numpy(golden.arr(m, "bo")))
    proc = SharedAttnProcessor(self_attn_idx=m["idx"] if m["N"] > 0 else None, save_self_attentions=m["save_probs"],
                               use_adain=m["use_adain"], train_input=m["train_input"])
    attn.set_processor(proc)
    hidden = torch.from_numpy(golden.arr(m, "hidden"))
    enc = golden.arr(m, "enc")
    enc = None if enc is None else torch.from_numpy(enc)
    kwargs = {"ref_keys": None, "ref_values": None}
    if m["N"] > 0:
        kwargs = {"ref_keys": [None] * m["idx"] + [torch.from_numpy(golden.arr(m, "ref_k"))],
                  "ref_values": [None] * m["idx"] + [torch.from_numpy(golden.arr(m, "ref_v"))]}
    with torch.no_grad():
        out = attn(hidden, encoder_hidden_states=enc, **kwargs)
    ref = golden.arr(m, "out")
    assert np.abs(out.numpy() - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max())
    if m["save_probs"]:
        assert np.abs(proc.attention_probs.numpy() - golden.arr(m, "probs")).max() <= 3e-5
    used = [c for c in shim.CALLS if c[0] == "shared_attention"][-1][1]
    assert used["include_self"] == (m["train_input"] if m["N"] > 0 else True)
    assert used["adain"] == (m["use_adain"] and m["N"] > 0)


def test_four_dim_input_and_residual_paths(shim):
    """the (B, C, H, W) entry form and residual / rescale knobs of the diffusers contract"""
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    attn = Attention(query_dim=64, heads=1, dim_head=64, processor=SharedAttnProcessor())
    x = torch.randn(2, 64, 4, 4)
    with torch.no_grad():
        y4 = attn(x)
        y3 = attn(x.view(2, 64, 16).transpose(1, 2))
        assert y4.shape == x.shape
        torch.testing.assert_close(y4, y3.transpose(1, 2).reshape(2, 64, 4, 4))
        attn.residual_connection, attn.rescale_output_factor = True, 2.0
        torch.testing.assert_close(attn(x), (y4 + x) / 2.0)
    with pytest.raises(NotImplementedError):
        attn(x, attention_mask=torch.zeros(2, 16, 16))


def test_adain_function_host_side(shim, golden):
    from face_replace.models.attn_processors import adain
    for m in (m for m in GOLDEN_MANIFEST if m["kind"] == "adain"):
        content, style = torch.from_numpy(golden.arr(m, "content")), torch.from_numpy(golden.arr(m, "style"))
        out = adain(content, style.mean(dim=1, keepdim=True), style.std(dim=1, keepdim=True) + 1e-5)
        ref = golden.arr(m, "out")
        assert np.abs(out.numpy() - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())


def test_product_has_no_cpu_fallback():
    """the real ops refuse CPU tensors and non-16-bit dtypes instead of silently computing"""
    from instantrestore_amd import ops
    from face_replace.models.attn_processors import SharedAttnProcessor
    from instantrestore_amd.attention import Attention
    q = torch.randn(1, 8, 64)
    for fn in (lambda: ops.shared_attention(q, q, q, heads=1, scale=0.125),
               lambda: ops.adain_stats(q, q[:, None], heads=1),
               lambda: ops.token_stats(q[:, None], heads=1)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fn()
    attn = Attention(query_dim=64, heads=1, dim_head=64, processor=SharedAttnProcessor())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        attn(q)
    # the product modules never import the oracle
    import instantrestore_amd, pkgutil, importlib, sys
    for mod in pkgutil.iter_modules(instantrestore_amd.__path__):
        if mod.name.startswith("lib"):  # the HIP shared object is not a Python extension module
            continue
        importlib.import_module(f"instantrestore_amd.{mod.name}")
    import inspect
    for name, module in list(sys.modules.items()):
        if name.startswith("instantrestore_amd") or name.startswith("face_replace"):
            src = inspect.getsource(module) if hasattr(module, "__file__") and module.__file__ else ""
            assert "import oracle" not in src and "from oracle" not in src, name


def test_reference_unet_early_exit_harvests_identical_kv(shim):
    """the reference UNet's output is discarded by the inference caller, so its forward
    may stop at the last K/V-capturing layer; the harvested lists must not change and nothing after that
    layer may run."""
    from face_replace.models.attn_processors import AttnProcessor, register_attention_processor_kv_unet
    from instantrestore_amd.kv_harvest import get_conditioning_keys_values
    from instantrestore_amd.unet_host import AttnTopologyUNet
    import __graft_entry__ as ge
    torch.manual_seed(0)
    unet = AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2),
                            cross_attention_dim=64, seed=5)
    ge.register_attention_processor_kv_unet_default(unet, CFG)
    register_attention_processor_kv_unet(unet)
    x = torch.randn(4, 4, 16, 16)
    text = torch.randn(4, 77, 64)
    with torch.no_grad():
        k_full, v_full = get_conditioning_keys_values(unet, x, None, text, 2, [2, 1])
        n_full = len([c for c in shim.CALLS if c[0] == "shared_attention"])
        shim.CALLS.clear()
        k_early, v_early = get_conditioning_keys_values(unet, x, None, text, 2, [2, 1], early_exit=True)
        n_early = len([c for c in shim.CALLS if c[0] == "shared_attention"])
    assert len(k_full) == len(k_early) == 9
    for a, b in zip(k_full + v_full, k_early + v_early):
        assert a.shape == b.shape and torch.equal(a, b)
    assert n_early < n_full                      # the last capture layer's attention and everything after it never ran
    procs = [p for p in unet.attn_processors.values() if type(p) in [AttnProcessor]]
    assert all(p.keys is None and p.stop_after_capture is None for p in procs)   # reset and disarmed
    assert len(procs[-1].state_dict()) == 0


def test_lora_cache_invalidation_api_and_load_state_dict_hook():
    """writes through .data do not bump _version - explicit invalidation and a load_state_dict post hook
    (installed by the registration functions) drop the folded weights; CPU-only logic test on the cache itself"""
    import torch
    from torch import nn
    from instantrestore_amd import lora_fold

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b = nn.Linear(8, 8, bias=False), nn.Linear(8, 8, bias=False)

    h = Holder()
    w1 = lora_fold.cached_weight(h, "_ir_qkv_cache", (h.a, h.b), torch.float32)
    assert w1.shape == (16, 8) and lora_fold.cached_weight(h, "_ir_qkv_cache", (h.a, h.b), torch.float32) is w1
    with torch.no_grad():
        h.a.weight.data.mul_(2.0)                          # invisible to the (data_ptr, _version) key
    assert lora_fold.cached_weight(h, "_ir_qkv_cache", (h.a, h.b), torch.float32) is w1
    lora_fold.invalidate(h)
    w2 = lora_fold.cached_weight(h, "_ir_qkv_cache", (h.a, h.b), torch.float32)
    assert w2 is not w1 and torch.equal(w2[:8], h.a.weight)
    with torch.no_grad():
        h.b.weight.mul_(3.0)                               # a normal in-place op bumps _version: refolded on its own
    assert torch.equal(lora_fold.cached_weight(h, "_ir_qkv_cache", (h.a, h.b), torch.float32)[8:], h.b.weight)
    outer = nn.Sequential(h)
    lora_fold.install_invalidation_hook(outer)
    lora_fold.install_invalidation_hook(outer)             # idempotent
    assert "_ir_qkv_cache" in h.__dict__
    outer.load_state_dict(outer.state_dict())
    assert "_ir_qkv_cache" not in h.__dict__
    assert lora_fold.invalidate_all(outer) == 0


def test_kv_cache_entries_are_compact_copies():
    """a cache entry must not be a view into the harvest's fused projection buffer"""
    import torch
    from instantrestore_amd.kv_cache import ReferenceKVCache
    fused = torch.randn(2 * 3, 10, 3 * 64)                              # (B*N, L, 3C) like the capture layer's q/k/v GEMM output
    k = fused[..., 64:128].reshape(2, 3, 10, 64)[:1]                     # strided view of identity 0
    v = fused[..., 128:].reshape(2, 3, 10, 64)[:1]
    cache = ReferenceKVCache(max_identities=2)
    ks, vs = cache.get_or_compute("a", lambda: ([k], [v]))
    assert ks[0].is_contiguous() and ks[0].untyped_storage().nbytes() == ks[0].numel() * 4
    assert torch.equal(ks[0], k) and torch.equal(vs[0], v)
    fused.zero_()                                                         # the producer's buffer is reused: the entry must not change
    assert float(ks[0].abs().max()) > 0
    assert cache.nbytes("a") == 2 * 3 * 10 * 64 * 4
    cache.get_or_compute("b", lambda: ([k], [v])); cache.get_or_compute("c", lambda: ([k], [v]))
    assert "a" not in cache and len(cache) == 2                           # LRU eviction


def test_registered_unet_still_pickles():
    """the load_state_dict post hook installed by the registration functions is a module-level function and the
    module keeps a flag, not the RemovableHandle - ``torch.save(unet)`` / multiprocessing spawn keep working"""
    import io
    from types import SimpleNamespace
    from instantrestore_amd.attn_processors import register_attention_processor
    from instantrestore_amd.unet_host import AttnTopologyUNet
    u = AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2), cross_attention_dim=64, seed=1)
    register_attention_processor(u, SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False))
    buf = io.BytesIO()
    torch.save(u, buf)
    assert buf.tell() > 0
    u.load_state_dict(u.state_dict())      # the hook still fires and returns None
