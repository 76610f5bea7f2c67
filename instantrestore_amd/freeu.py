"""FreeU on the skip connections of the first two up blocks, MI355X-native.

Same call contract as the reference's ``apply_freeu`` (``face_replace/models/unet_2d_condition/
block.py:3495-3520``, enabled with ``s1=0.9, s2=0.2, b1=1.4, b2=1.6`` at ``pix2pix_turbo.py:62-68``)
and as the ``fourier_filter`` it calls (diffusers==0.24.0 ``utils/torch_utils.py``).  The reference
spells the filter as ``.float()`` -> ``fftn`` -> ``fftshift`` -> ones/mask -> multiply ->
``ifftshift`` -> ``ifftn`` -> ``.real`` -> ``.to(dtype)``: ten launches and several fp32 round trips
through HBM for tensors of a few MB.  Here it is one HIP kernel that reads the 16-bit tensor once
and writes it once (``ir_freeu_fourier_filter``).  No torch fallback.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import ops as _ops


def fourier_filter(x_in: torch.Tensor, threshold: int, scale: float) -> torch.Tensor:
    """``fourier_filter(x_in, threshold, scale)``: returns a tensor of ``x_in``'s dtype."""
    return _ops.freeu_fourier_filter(x_in, threshold, scale)


def apply_freeu(resolution_idx: int, hidden_states: torch.Tensor, res_hidden_states: torch.Tensor,
                **freeu_kwargs) -> Tuple[torch.Tensor, torch.Tensor]:
    """Backbone gain on the first half of the channels (in place, like the reference) and the
    Fourier filter on the skip features, for ``resolution_idx`` 0 and 1; identity otherwise."""
    if resolution_idx == 0:
        gain, s = freeu_kwargs["b1"], freeu_kwargs["s1"]
    elif resolution_idx == 1:
        gain, s = freeu_kwargs["b2"], freeu_kwargs["s2"]
    else:
        return hidden_states, res_hidden_states
    half = hidden_states.shape[1] // 2
    hidden_states[:, :half].mul_(gain)   # block.py:3513,3517: x[:, :half] = x[:, :half] * b
    res_hidden_states = _ops.freeu_fourier_filter(res_hidden_states, 1, s)
    return hidden_states, res_hidden_states
