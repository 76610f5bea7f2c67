"""Inference-time folding of LoRA adapters into the projection weights.

The reference wraps ``to_q`` / ``to_k`` / ``to_v`` / ``to_out.0`` of the main UNet with peft
LoRA layers (``pix2pix_turbo.py:171-179``, ``peft==0.10.0`` per ``environment_new.yml:234``):
every projection is then three GEMMs and two elementwise kernels
(``base(x) + B(A(dropout(x))) * scaling``).  At inference the adapter is constant, so the same
linear map is one GEMM against ``W + scaling * B @ A`` - what peft's own ``merge_and_unload``
computes.  Nothing is written back to the modules (checkpoints keep loading ``strict=True``,
test.py:47-50): the folded weight lives in a per-module cache keyed on the identity and
``_version`` of every tensor that went into it, so optimiser steps or ``load_state_dict``
invalidate it.  Writes through ``.data`` (``w.data.copy_``, ``w.data -= d``: diffusers'
``EMAModel.copy_to``, manual weight surgery) do NOT bump ``_version``: call :func:`invalidate` /
:func:`invalidate_all` after them.  The registration functions and a ``load_state_dict`` post hook
installed by them do so on their own.

peft is not installed in this image; the wrapper is recognised structurally
(``base_layer`` / ``lora_A`` / ``lora_B`` / ``scaling`` / ``active_adapters``), which is the
layout of ``peft.tuners.lora.layer.Linear`` in that release.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import nn


def _is_identity_dropout(mod, training: bool) -> bool:
    if mod is None or isinstance(mod, nn.Identity):
        return True
    if isinstance(mod, nn.Dropout):
        return (not training) or mod.p == 0.0
    return False


def _lora_parts(mod):
    """(base nn.Linear, [(A weight, B weight, scaling)]) of a peft-style LoRA Linear, or None when
    ``mod`` is not one or is in a state whose forward is not a plain linear map."""
    base = getattr(mod, "base_layer", None)
    if not isinstance(base, nn.Linear) or not hasattr(mod, "lora_A") or not hasattr(mod, "lora_B"):
        return None
    if getattr(mod, "fan_in_fan_out", False):
        return None
    if getattr(mod, "disable_adapters", False) or getattr(mod, "merged", False):
        return base, []  # peft then calls the base layer alone (weights already merged, or adapters off)
    active = getattr(mod, "active_adapters", None)
    if active is None:
        active = getattr(mod, "active_adapter", [])
    if isinstance(active, str):
        active = [active]
    parts = []
    for name in active:
        if name not in mod.lora_A:
            continue
        dora = getattr(mod, "use_dora", None)
        if isinstance(dora, dict) and dora.get(name, False):
            return None  # DoRA renormalises per output column: not a constant additive delta
        drop = mod.lora_dropout[name] if hasattr(mod, "lora_dropout") and name in mod.lora_dropout else None
        if not _is_identity_dropout(drop, mod.training):
            return None
        a, b = mod.lora_A[name], mod.lora_B[name]
        if getattr(a, "bias", None) is not None or getattr(b, "bias", None) is not None:
            return None
        parts.append((a.weight, b.weight, float(mod.scaling[name])))
    return base, parts


def effective_linear(mod) -> Optional[Tuple[nn.Linear, list]]:
    """``(base, lora parts)`` when ``mod(x)`` equals ``F.linear(x, W_eff, base.bias)`` for a constant
    ``W_eff``: a plain ``nn.Linear`` or a foldable LoRA wrapper.  ``None`` otherwise."""
    if type(mod) is nn.Linear:
        return mod, []
    return _lora_parts(mod)


def _key(base: nn.Linear, parts, dtype, extra=()):
    k = [base.weight.data_ptr(), base.weight._version, base.weight.device, dtype]
    for a, b, s in parts:
        k += [a.data_ptr(), a._version, b.data_ptr(), b._version, s]
    return tuple(k) + tuple(extra)


def folded_weight(base: nn.Linear, parts, dtype) -> torch.Tensor:
    """``W + sum_i scaling_i * B_i @ A_i`` accumulated in fp32, rounded once to ``dtype``."""
    w = base.weight.detach()
    if not parts:
        return w.to(dtype)
    acc = w.float()
    for a, b, s in parts:
        acc = acc + (b.detach().float() @ a.detach().float()) * s
    return acc.to(dtype)


def cached_weight(owner, slot: str, mods, dtype) -> Optional[torch.Tensor]:
    """Folded (and, for several modules, row-concatenated) weight of ``mods`` cached on ``owner``
    (a plain attribute: never part of a state dict).  ``None`` if any module is not foldable."""
    effs = []
    for m in mods:
        e = effective_linear(m)
        if e is None:
            return None
        effs.append(e)
    key = tuple(_key(base, parts, dtype) for base, parts in effs)
    cache = owner.__dict__.get(slot)
    if cache is None or cache[0] != key:
        ws = [folded_weight(base, parts, dtype) for base, parts in effs]
        w = ws[0] if len(ws) == 1 else torch.cat(ws, dim=0)
        cache = (key, w.contiguous())
        owner.__dict__[slot] = cache
    return cache[1]


def cached_cast(owner, slot: str, t: torch.Tensor, dtype) -> torch.Tensor:
    """``t.to(dtype)`` cached on ``owner`` (bias under autocast), keyed on identity and version"""
    key = (t.data_ptr(), t._version, t.device, dtype)
    cache = owner.__dict__.get(slot)
    if cache is None or cache[0] != key:
        cache = (key, t.detach().to(dtype))
        owner.__dict__[slot] = cache
    return cache[1]


_SLOTS = ("_ir_qkv_cache", "_ir_out_cache", "_ir_out_bias_cache", "_ir_q_cache", "_ir_q_bias_cache", "_ir_kv_cache")


def invalidate(attn) -> None:
    """drop the folded / fused weights cached on one attention module (they are rebuilt on its next call)"""
    for slot in _SLOTS:
        attn.__dict__.pop(slot, None)


def invalidate_all(module: nn.Module) -> int:
    """:func:`invalidate` on every sub-module of ``module`` that carries a cache; returns how many did"""
    n = 0
    for m in module.modules():
        if any(slot in m.__dict__ for slot in _SLOTS):
            invalidate(m)
            n += 1
    return n


def _invalidate_after_load(mod, incompatible_keys) -> None:
    """``load_state_dict`` post hook (module-level, so a UNet carrying it still pickles: ``torch.save(unet)``,
    multiprocessing spawn); a post hook must return None (torch asserts it)"""
    invalidate_all(mod)


def install_invalidation_hook(module: nn.Module) -> None:
    """``load_state_dict`` on ``module`` (checkpoint load, test.py:47-50) drops every cached folded weight below it.
    Only a flag is kept on the module (the ``RemovableHandle`` holds weak references that do not pickle)."""
    if not getattr(module, "_ir_invalidation_hook", False) and hasattr(module, "register_load_state_dict_post_hook"):
        module.register_load_state_dict_post_hook(_invalidate_after_load)
        module._ir_invalidation_hook = True
