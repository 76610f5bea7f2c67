"""Minimal attention *host* module: the object an attention processor receives as ``attn``.

The reference never defines this class itself: it is ``diffusers==0.24.0``'s ``Attention``
(``environment_new.yml:89``), which is not vendored under the reference tree and is not
installed here.  This file is a from-scratch statement of the part of that contract that the
processors of ``face_replace/models/attn_processors.py:44-95,205-277`` touch (SURVEY.md
Appendix A): the projection layers, the head split / merge helpers, the score helper and the
``processor`` plug point.  It is host plumbing (plain torch ``nn.Linear``), not the hot path:
the fused HIP kernels are reached from the processors, never from here.

``get_attention_scores`` exists because third-party processors (and the reference's own, when
they are driven through this host to produce golden vectors) call it; the processors of this
package do not.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


class Attention(nn.Module):
    """Projection container + helper methods, diffusers-0.24 flavoured.

    Parameters mirror the subset of the upstream constructor that SD-Turbo uses:
    ``query_dim``, ``cross_attention_dim`` (``None`` => self attention), ``heads``,
    ``dim_head`` (64 everywhere in SD-Turbo), bias-free q/k/v, biased out projection,
    ``Dropout(0)``.
    """

    def __init__(
        self,
        query_dim: int,
        cross_attention_dim: Optional[int] = None,
        heads: int = 8,
        dim_head: int = 64,
        dropout: float = 0.0,
        bias: bool = False,
        out_bias: bool = True,
        upcast_attention: bool = False,
        upcast_softmax: bool = False,
        processor: Optional[nn.Module] = None,
    ) -> None:
        super().__init__()
        inner = heads * dim_head
        kv_dim = query_dim if cross_attention_dim is None else cross_attention_dim
        self.heads = heads
        self.inner_dim = inner
        self.scale = dim_head ** -0.5
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        # knobs the processors probe; all inert inside the UNet's transformer blocks
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0

        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(kv_dim, inner, bias=bias)
        self.to_v = nn.Linear(kv_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(dropout)])

        self.processor: Optional[nn.Module] = None
        if processor is not None:
            self.set_processor(processor)

    # ---- plug point ------------------------------------------------------------------
    def set_processor(self, processor, _remove_lora: bool = False) -> None:
        # an nn.Module processor becomes the sub-module "processor" (so .to() reaches it and
        # its - empty - state dict is part of the model's)
        if isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor", None)
        self.processor = processor

    def get_processor(self, return_deprecated_lora: bool = False):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **cross_attention_kwargs):
        return self.processor(
            self,
            hidden_states,
            encoder_hidden_states=encoder_hidden_states,
            attention_mask=attention_mask,
            **cross_attention_kwargs,
        )

    # ---- helpers ---------------------------------------------------------------------
    def head_to_batch_dim(self, t: torch.Tensor) -> torch.Tensor:
        b, s, c = t.shape
        h = self.heads
        return t.reshape(b, s, h, c // h).permute(0, 2, 1, 3).reshape(b * h, s, c // h)

    def batch_to_head_dim(self, t: torch.Tensor) -> torch.Tensor:
        bh, s, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            base = torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device)
            beta = 0
        else:
            base, beta = attention_mask, 1
        scores = torch.baddbmm(base, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        return scores.softmax(dim=-1).to(dtype)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size, out_dim: int = 3):
        if attention_mask is None:
            return None
        raise NotImplementedError(
            "attention masks never occur on the InstantRestore path (pix2pix_turbo.py:317-326 "
            "calls the UNet without one); this host does not prepare them"
        )
