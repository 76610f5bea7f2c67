"""Work-per-unit arithmetic of the hot path.

Algorithmic figures only - QK^T and PV contractions for FLOPs; each of Q, K, V read once and O
written once for bytes - so that ``achieved / peak`` means the same thing for every kernel
variant.  Peaks are the MI355X numbers of ``/opt/skills/guides/MI355X_MICROARCH.md``.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

MFMA_PEAK_TFLOPS_16BIT = 2500.0  # dense bf16/fp16 (the marketing 5 PF figure is 2:1 sparse)
HBM_PEAK_GBS = 8000.0            # spec; ~6300 achievable

# (L, C, H) of the three decoder layer classes at 512 px, three layers each
# (face_replace/data/datasets/restore_dataset.py:71-75)
LAYER_CLASSES_512: List[Tuple[int, int, int]] = [(256, 1280, 20), (1024, 640, 10), (4096, 320, 5)]


def layer_classes(px: int = 512) -> List[Tuple[int, int, int]]:
    f = (px // 512) ** 2
    return [(L * f, C, H) for (L, C, H) in LAYER_CLASSES_512]


def attn_flops(batch: int, lq: int, lkv: int, channels: int) -> float:
    """4 * B * Lq * Lkv * C  (2 for QK^T + 2 for PV; softmax / AdaIN excluded)"""
    return 4.0 * batch * lq * lkv * channels


def attn_bytes(batch: int, lq: int, lkv: int, channels: int, elem: int = 2) -> float:
    """read Q, write O, read K and V once each"""
    return float(elem) * batch * channels * (2 * lq + 2 * lkv)


def adain_stats_bytes(batch: int, l_self: int, n_refs: int, l_ref: int, channels: int, elem: int = 2) -> float:
    """folded AdaIN: one read of V_self and of every reference V"""
    return float(elem) * batch * channels * (l_self + n_refs * l_ref)


def shared_path_flops_per_identity(n_refs: int, train_input: bool, px: int = 512) -> float:
    t = 1 if train_input else 0
    return sum(3 * attn_flops(1, L, (n_refs + t) * L, C) for (L, C, _) in layer_classes(px))


def kv_capture_flops_per_identity(n_refs: int, px: int = 512) -> float:
    """plain self-attention of the frozen reference UNet's nine decoder layers over N references"""
    return sum(3 * attn_flops(n_refs, L, L, C) for (L, C, _) in layer_classes(px))


def summary(n_refs: int, train_input: bool, px: int = 512) -> Dict[str, float]:
    return {
        "shared_gflop_per_identity": shared_path_flops_per_identity(n_refs, train_input, px) / 1e9,
        "kv_capture_gflop_per_identity": kv_capture_flops_per_identity(n_refs, px) / 1e9,
    }
