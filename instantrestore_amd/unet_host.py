"""Attention-topology host: the plugin surface of the reference's UNet, without the UNet.

The reference's UNet host (``face_replace/models/unet_2d_condition/unet.py``, a vendored
diffusers-0.24 ``UNet2DConditionModel``) is conv/GEMM-bound upstream code and is OUT OF SCOPE of
this build; what the hot path needs from it is its *plugin surface*:

* ``attn_processors``  -> ``{"<module path>.processor": processor}`` in registration order
  (unet.py:628-650),
* ``set_attn_processor(dict)`` with the length check and pop-by-name semantics (unet.py:652-686),
* ``config.cross_attention_dim`` / ``config.block_out_channels``, ``device`` / ``dtype``,
* module names that start with ``down_blocks.<i>`` / ``up_blocks.<i>`` / ``mid_block`` and end in
  ``attn1.processor`` / ``attn2.processor`` (the registration functions parse them textually),
* ``forward(sample, timestep, encoder_hidden_states, cross_attention_kwargs={'ref_keys': ...,
  'ref_values': ...}).sample`` handing the kwargs to every attention (pix2pix_turbo.py:323-326).

:class:`AttnTopologyUNet` reproduces exactly that, with SD-Turbo's attention topology
(16 transformer blocks = 32 processors; decoder self-attentions at 16x16 / 32x32 / 64x64 tokens
with 1280/640/320 channels = 20/10/5 heads of 64; restore_dataset.py:71-75) and seeded synthetic
weights - no checkpoint is reachable offline.  Between attention stages the activations are moved
with cheap token-space stand-ins (2x2 mean pooling / nearest up-sampling + a ``Linear``) where the
real UNet has ResNet/conv blocks; they exist only so every attention sees a tensor of the right
shape.  Tests, ``smoke()``, ``examples/synthetic_inference.py`` and ``bench.py --e2e`` use it.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
from torch import nn

from .attention import Attention

SD_TURBO = dict(
    block_out_channels=(320, 640, 1280, 1280),
    attention_head_dim=(5, 10, 20, 20),  # = number of heads; head dim is 64 everywhere
    cross_attention_dim=1024,
    layers_per_block=2,
    sample_size=64,
    in_channels=4,
)


class TransformerBlock(nn.Module):
    """norm1 -> attn1 (self) -> norm2 -> attn2 (cross) with residuals (diffusers
    BasicTransformerBlock minus the feed-forward, which is GEMM work outside the path)."""

    def __init__(self, dim: int, heads: int, cross_dim: int):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(query_dim=dim, heads=heads, dim_head=dim // heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_dim, heads=heads, dim_head=dim // heads)

    def forward(self, h, encoder_hidden_states, kwargs):
        h = h + self.attn1(self.norm1(h), **kwargs)
        h = h + self.attn2(self.norm2(h), encoder_hidden_states=encoder_hidden_states, **kwargs)
        return h


class Transformer2D(nn.Module):
    def __init__(self, dim: int, heads: int, cross_dim: int):
        super().__init__()
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([TransformerBlock(dim, heads, cross_dim)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, h, encoder_hidden_states, kwargs):
        res = h
        h = self.proj_in(h)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states, kwargs)
        return self.proj_out(h) + res


class AttnStage(nn.Module):
    """``n_attn`` transformer modules at one resolution (``attentions``) preceded by the
    stand-in that brings tokens to this stage's channel width."""

    def __init__(self, c_in: int, dim: int, heads: int, cross_dim: int, n_attn: int):
        super().__init__()
        self.lift = nn.Linear(c_in, dim)
        self.attentions = nn.ModuleList([Transformer2D(dim, heads, cross_dim) for _ in range(n_attn)])

    def forward(self, h, encoder_hidden_states, kwargs):
        h = self.lift(h)
        for a in self.attentions:
            h = a(h, encoder_hidden_states, kwargs)
        return h


def _pool2(h: torch.Tensor) -> torch.Tensor:
    b, l, c = h.shape
    s = int(round(l ** 0.5))
    return h.reshape(b, s // 2, 2, s // 2, 2, c).mean(dim=(2, 4)).reshape(b, (s // 2) ** 2, c)


def _up2(h: torch.Tensor) -> torch.Tensor:
    b, l, c = h.shape
    s = int(round(l ** 0.5))
    return h.reshape(b, s, 1, s, 1, c).expand(b, s, 2, s, 2, c).reshape(b, 4 * l, c)


class AttnTopologyUNet(nn.Module):
    """See the module docstring.  ``sample`` is a latent ``(B, in_channels, S, S)``."""

    def __init__(self, block_out_channels=SD_TURBO["block_out_channels"],
                 attention_head_dim=SD_TURBO["attention_head_dim"],
                 cross_attention_dim: int = SD_TURBO["cross_attention_dim"],
                 layers_per_block: int = SD_TURBO["layers_per_block"],
                 in_channels: int = SD_TURBO["in_channels"], seed: Optional[int] = 0):
        super().__init__()
        boc, heads = tuple(block_out_channels), tuple(attention_head_dim)
        self.config = SimpleNamespace(block_out_channels=boc, attention_head_dim=heads,
                                      cross_attention_dim=cross_attention_dim,
                                      layers_per_block=layers_per_block, in_channels=in_channels)
        if seed is not None:
            rng_state = torch.random.get_rng_state()
            torch.manual_seed(seed)
        nb = len(boc)
        # registration order matters: down_blocks, up_blocks, mid_block (unet.py:416-417,488)
        self.conv_in = nn.Linear(in_channels, boc[0])
        self.down_blocks = nn.ModuleList()
        c_prev = boc[0]
        for i in range(nb):
            n_attn = layers_per_block if i < nb - 1 else 0  # last down block has no attention
            self.down_blocks.append(AttnStage(c_prev, boc[i], heads[i], cross_attention_dim, n_attn))
            c_prev = boc[i]
        self.up_blocks = nn.ModuleList()
        rev, rheads = list(reversed(boc)), list(reversed(heads))
        for i in range(nb):
            n_attn = (layers_per_block + 1) if i > 0 else 0  # first up block has no attention
            self.up_blocks.append(AttnStage(c_prev, rev[i], rheads[i], cross_attention_dim, n_attn))
            c_prev = rev[i]
        self.mid_block = AttnStage(boc[-1], boc[-1], heads[-1], cross_attention_dim, 1)
        self.conv_out = nn.Linear(boc[0], in_channels)
        if seed is not None:
            torch.random.set_rng_state(rng_state)

    # ---- plugin surface (unet.py:628-686) -------------------------------------------------
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @property
    def dtype(self) -> torch.dtype:
        return next(self.parameters()).dtype

    @property
    def attn_processors(self) -> Dict[str, nn.Module]:
        found: Dict[str, nn.Module] = {}

        def visit(name: str, module: nn.Module):
            if hasattr(module, "get_processor"):
                found[f"{name}.processor"] = module.get_processor(return_deprecated_lora=True)
            for child_name, child in module.named_children():
                visit(f"{name}.{child_name}", child)

        for name, module in self.named_children():
            visit(name, module)
        return found

    def set_attn_processor(self, processor, _remove_lora: bool = False) -> None:
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(f"set_attn_processor: got a dict with {len(processor)} entries for {count} attention layers "
                             "(one processor per layer, keyed '<module path>.processor')")

        def visit(name: str, module: nn.Module):
            if hasattr(module, "set_processor"):
                if isinstance(processor, dict):
                    module.set_processor(processor.pop(f"{name}.processor"), _remove_lora=_remove_lora)
                else:
                    module.set_processor(processor, _remove_lora=_remove_lora)
            for child_name, child in module.named_children():
                visit(f"{name}.{child_name}", child)

        for name, module in self.named_children():
            visit(name, module)

    def decoder_self_attentions(self) -> List[Attention]:
        """the nine ``up_blocks.*attn1`` modules in ``self_attn_idx`` order"""
        return [m for n, m in self.named_modules()
                if n.startswith("up_blocks") and n.endswith("attn1") and isinstance(m, Attention)]

    # ---- forward -----------------------------------------------------------------------
    def forward(self, sample, timestep=None, encoder_hidden_states=None, cross_attention_kwargs=None):
        kwargs = dict(cross_attention_kwargs or {})
        b, c, hh, ww = sample.shape
        h = self.conv_in(sample.reshape(b, c, hh * ww).transpose(1, 2))
        nb = len(self.down_blocks)
        for i, blk in enumerate(self.down_blocks):
            h = blk(h, encoder_hidden_states, kwargs)
            if i < nb - 1:
                h = _pool2(h)
        h = self.mid_block(h, encoder_hidden_states, kwargs)
        for i, blk in enumerate(self.up_blocks):
            h = blk(h, encoder_hidden_states, kwargs)
            if i < nb - 1:
                h = _up2(h)
        out = self.conv_out(h).transpose(1, 2).reshape(b, c, hh, ww)
        return SimpleNamespace(sample=out)
