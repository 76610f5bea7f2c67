#!/usr/bin/env python3
"""Pin the third-party seam the day the packages exist.

The hot path is written against three pieces of third-party code that are neither in /root/reference nor in this image, so the
repo ships stand-ins restated from the published release and says "unpinned" wherever they matter:

  1. diffusers == 0.24.* `Attention`           -> instantrestore_amd/attention.py          (helpers the processors call)
  2. peft LoRA wrappers on to_q / k / v / out  -> instantrestore_amd/lora_fold.py           (`effective_linear`, `folded_weight`)
  3. `UNet2DConditionModel.attn_processors` /   -> instantrestore_amd/unet_host.py           (key order and the
     `set_attn_processor`                                                                    "<path>.processor" naming; the
     reference's vendored copy: face_replace/models/unet_2d_condition/unet.py:628-686)

Run this script - or `pytest tests/test_third_party_seam.py` - in an environment that has the real packages: every check
compares the real object with the stand-in on the same inputs and fails loudly on the first difference.  Without the packages
each check reports "skipped: <package> not importable" and the exit code is 0 (nothing was pinned, nothing is claimed).
CPU only: no GPU and none of this library's kernels are involved - the seam is host-side arithmetic and naming."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _have(name):
    try:
        return importlib.import_module(name)
    except Exception:
        return None


def check_attention_contract():
    """real diffusers `Attention` against the stand-in: attributes the processors read, head split / merge, attention scores
    (baddbmm with alpha = scale -> softmax), mask preparation for mask = None, processor registration"""
    diffusers = _have("diffusers")
    if diffusers is None:
        return "skipped: diffusers not importable"
    ver = getattr(diffusers, "__version__", "?")
    from diffusers.models.attention_processor import Attention as Real
    from instantrestore_amd.attention import Attention as Ours
    torch.manual_seed(0)
    for heads, cross in ((5, None), (10, 1024), (20, None)):
        dim = heads * 64
        real = Real(query_dim=dim, cross_attention_dim=cross, heads=heads, dim_head=64)
        ours = Ours(query_dim=dim, cross_attention_dim=cross, heads=heads, dim_head=64)
        assert real.heads == ours.heads and abs(real.scale - ours.scale) < 1e-12, "heads / scale"
        for name in ("to_q", "to_k", "to_v"):
            a, b = getattr(real, name), getattr(ours, name)
            assert tuple(a.weight.shape) == tuple(b.weight.shape) and (a.bias is None) == (b.bias is None), name
        assert tuple(real.to_out[0].weight.shape) == tuple(ours.to_out[0].weight.shape) and real.to_out[0].bias is not None
        assert len(real.to_out) == len(ours.to_out) == 2
        for attr in ("residual_connection", "rescale_output_factor", "upcast_attention", "upcast_softmax"):
            assert getattr(real, attr) == getattr(ours, attr), attr
        assert real.spatial_norm is None and real.group_norm is None and real.norm_cross is None
        x = torch.randn(2, 37, dim)
        assert torch.equal(real.head_to_batch_dim(x), ours.head_to_batch_dim(x)), "head_to_batch_dim"
        y = torch.randn(2 * heads, 37, 64)
        assert torch.equal(real.batch_to_head_dim(y), ours.batch_to_head_dim(y)), "batch_to_head_dim"
        q, k = torch.randn(2 * heads, 37, 64), torch.randn(2 * heads, 53, 64)
        assert torch.allclose(real.get_attention_scores(q, k), ours.get_attention_scores(q, k), atol=1e-6), "get_attention_scores"
        assert real.prepare_attention_mask(None, 53, 2) is None and ours.prepare_attention_mask(None, 53, 2) is None
        # a module processor is registered as sub-module `processor`; forward hands every cross_attention_kwarg on
        from face_replace.models.attn_processors import SharedAttnProcessor
        p = SharedAttnProcessor(self_attn_idx=None)
        real.set_processor(p)
        assert real.get_processor() is p and dict(real.named_modules())["processor"] is p
    return "ok (diffusers %s)" % ver


def check_lora_fold():
    """real peft LoRA wrapper through lora_fold: the folded weight reproduces the wrapper's own forward"""
    peft = _have("peft")
    if peft is None:
        return "skipped: peft not importable"
    from peft import LoraConfig, get_peft_model
    from instantrestore_amd import lora_fold
    torch.manual_seed(1)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.to_q = torch.nn.Linear(64, 64, bias=False)
            self.to_out = torch.nn.ModuleList([torch.nn.Linear(64, 64), torch.nn.Dropout(0.0)])

        def forward(self, x):
            return self.to_out[0](self.to_q(x))
    m = get_peft_model(Tiny(), LoraConfig(r=8, lora_alpha=8, init_lora_weights="gaussian", target_modules=["to_q", "to_out.0"]))
    for mod in m.modules():
        if hasattr(mod, "lora_B"):
            for lb in mod.lora_B.values():
                torch.nn.init.normal_(lb.weight, std=0.05)      # non-zero B: the adapter does something
    m.eval()
    inner = m.base_model.model
    x = torch.randn(3, 10, 64)
    for wrapped in (inner.to_q, inner.to_out[0]):
        got = lora_fold.effective_linear(wrapped)
        assert got is not None, "lora_fold does not recognise peft's wrapper %r" % type(wrapped)
        base, parts = got
        w = lora_fold.folded_weight(base, parts, torch.float32)
        want = wrapped(x)
        have = torch.nn.functional.linear(x, w, base.bias)
        assert torch.allclose(want, have, atol=1e-5), "folded weight != wrapper forward (%s)" % type(wrapped).__name__
    return "ok (peft %s)" % getattr(peft, "__version__", "?")


def check_unet_processor_keys():
    """real UNet2DConditionModel (SD-2.x topology, tiny widths): attn_processors key order and naming vs the stand-in host"""
    diffusers = _have("diffusers")
    if diffusers is None:
        return "skipped: diffusers not importable"
    from diffusers import UNet2DConditionModel
    from instantrestore_amd.unet_host import AttnTopologyUNet
    real = UNet2DConditionModel(sample_size=8, in_channels=4, out_channels=4, block_out_channels=(64, 128, 128, 128),
                                attention_head_dim=(1, 2, 2, 2), cross_attention_dim=64, layers_per_block=2,
                                down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
                                up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
                                use_linear_projection=True, norm_num_groups=32)
    ours = AttnTopologyUNet(block_out_channels=(64, 128, 128, 128), attention_head_dim=(1, 2, 2, 2), cross_attention_dim=64)
    rk, ok = list(real.attn_processors.keys()), list(ours.attn_processors.keys())
    assert rk == ok, "attn_processors keys differ:\n real %s\n ours %s" % (rk[:6], ok[:6])
    # registration through the reference-compatible function lands on the same names with the same indices
    from types import SimpleNamespace
    from face_replace.models.attn_processors import register_attention_processor
    cfg = SimpleNamespace(use_adain=True, train_input=True, condition_on_face_embeds=False)
    register_attention_processor(real, cfg)
    register_attention_processor(ours, cfg)
    a = {n: (type(p).__name__, p.self_attn_idx) for n, p in real.attn_processors.items()}
    b = {n: (type(p).__name__, p.self_attn_idx) for n, p in ours.attn_processors.items()}
    assert a == b, "registration maps differ"
    return "ok (%d processors, diffusers %s)" % (len(rk), getattr(diffusers, "__version__", "?"))


CHECKS = [("diffusers Attention contract", check_attention_contract), ("peft LoRA through lora_fold", check_lora_fold),
          ("UNet2DConditionModel processor keys", check_unet_processor_keys)]


def main():
    bad = 0
    for name, fn in CHECKS:
        try:
            res = fn()
        except AssertionError as e:
            res, bad = "FAILED: %s" % e, bad + 1
        print("%-40s %s" % (name, res))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
