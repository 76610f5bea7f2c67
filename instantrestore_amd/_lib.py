"""ctypes binding of ``libinstantrestore_hip.so`` (C ABI: ``include/instantrestore_hip.h``).

There is exactly one compute backend - the hand-written HIP library built by
``instantrestore_amd/csrc/build.sh`` - and no fallback: if the shared object is missing or
cannot be loaded, :func:`lib` raises ``ImportError`` and every op above it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# IR_LIB_PATH: load an alternative build of the same ABI (compiler-flag A/B experiments)
LIB_PATH = os.environ.get("IR_LIB_PATH") or os.path.join(_HERE, "libinstantrestore_hip.so")
ABI_VERSION = 9

IR_DTYPE_F16, IR_DTYPE_BF16 = 0, 1
IR_FLAG_INCLUDE_SELF, IR_FLAG_Q_PRESCALED, IR_FLAG_OUT_F32 = 1, 2, 4

i32, i64, f32, u32, vp = C.c_int32, C.c_int64, C.c_float, C.c_uint32, C.c_void_p


class SharedAttnArgs(C.Structure):
    """mirror of ``ir_shared_attn_args`` (field order and types must match the header)"""

    _fields_ = (
        [("struct_size", u32), ("dtype", i32), ("flags", u32), ("batch", i32), ("heads", i32),
         ("len_q", i32), ("len_self", i32), ("n_refs", i32), ("len_ref", i32), ("scale", f32)]
        + [(n, vp) for n in ("q", "k_self", "v_self", "k_ref", "v_ref", "adain_a", "adain_b", "out", "lse")]
        + [(n, i64) for n in ("q_sb", "q_sl", "q_sh", "ks_sb", "ks_sl", "ks_sh", "vs_sb", "vs_sl", "vs_sh",
                              "kr_sb", "kr_sn", "kr_sl", "kr_sh", "vr_sb", "vr_sn", "vr_sl", "vr_sh",
                              "o_sb", "o_sl", "o_sh")]
        + [("workspace", vp), ("workspace_bytes", C.c_uint64), ("tuning", i32), ("reserved", i32), ("valid_refs", vp), ("seg_mass", vp)]
    )


class ImageDesc(C.Structure):
    """mirror of ``ir_image_desc`` (one source image of ``ir_preprocess_lanczos_u8``)"""

    _fields_ = [("src", vp), ("src_row_bytes", i64), ("in_h", i32), ("in_w", i32), ("out_h", i32), ("out_w", i32),
                ("crop_top", i32), ("crop_left", i32), ("bounds_h", vp), ("kk_h", vp), ("bounds_v", vp), ("kk_v", vp),
                ("ksize_h", i32), ("ksize_v", i32), ("row_first", i32), ("row_count", i32),
                ("col_first", i32), ("col_count", i32), ("tmp", vp)]


# name -> (restype, argtypes); every symbol include/instantrestore_hip.h declares
SYMBOLS = {
    "ir_abi_version": (C.c_int, []),
    "ir_build_info": (C.c_char_p, []),
    "ir_last_error_string": (C.c_char_p, []),
    "ir_shared_attn_workspace_bytes": (C.c_size_t, []),
    "ir_shared_attn_fwd": (C.c_int, [C.POINTER(SharedAttnArgs), vp]),
    "ir_shared_attn_kernel_name": (C.c_char_p, [C.POINTER(SharedAttnArgs)]),
    "ir_time_shared_attn_fwd": (C.c_int, [C.POINTER(SharedAttnArgs), i32, vp, C.POINTER(f32)]),
    "ir_bench_mfma_stream_scratch_bytes": (C.c_size_t, []),
    "ir_bench_mfma_stream": (C.c_int, [i32, i32, i32, i32, vp, C.c_size_t, vp, C.POINTER(f32)]),
    "ir_attn_probs": (C.c_int, [C.POINTER(SharedAttnArgs), vp, vp]),
    "ir_attn_probs_ex": (C.c_int, [C.POINTER(SharedAttnArgs), vp, i32, vp]),
    "ir_attn_segment_mass": (C.c_int, [C.POINTER(SharedAttnArgs), vp, vp]),
    "ir_adain_stats_workspace_bytes": (C.c_size_t, [i32, i32, i32, i32, i32]),
    "ir_adain_stats": (C.c_int, [i32, i32, i32, i32, i32, i32, vp, i64, i64, i64, vp, i64, i64, i64, i64,
                                 f32, vp, vp, vp, C.c_size_t, vp]),
    "ir_adain_stats_cached": (C.c_int, [i32, i32, i32, i32, i32, vp, i64, i64, i64, vp, vp, f32, vp, vp, vp, C.c_size_t, vp]),
    "ir_token_stats": (C.c_int, [i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, vp, vp, vp, C.c_size_t, vp]),
    "ir_adain_apply": (C.c_int, [i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, vp, vp,
                                 vp, i64, i64, i64, i64, vp]),
    "ir_tensor2im_u8": (C.c_int, [i32, i32, i32, i32, i32, vp, i64, i64, i64, i64, vp, vp]),
    "ir_lanczos_ksize": (C.c_int, [i32, i32]),
    "ir_lanczos_coeffs": (C.c_int, [i32, i32, vp, vp]),
    "ir_preprocess_lanczos_u8": (C.c_int, [C.POINTER(ImageDesc), i32, i32, i32, vp, vp]),
    "ir_freeu_fourier_filter": (C.c_int, [i32, i64, i32, i32, vp, i64, vp, i64, i32, f32, vp]),
    "ir_linear_fwd": (C.c_int, [i32, i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, vp]),
    "ir_linear_fwd_scaled": (C.c_int, [i32, i32, i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, i32, f32, vp]),
    "ir_linear_fwd_ex": (C.c_int, [i32, i32, i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, i32, f32, i32, vp]),
    "ir_linear_kernel_for": (C.c_int, [i64, i32, i32, i32]),
    "ir_linear_stats_rows": (C.c_int, [i64, i32, i32, i32]),
    "ir_linear_fwd_stats": (C.c_int, [i32, i32, i64, i32, i32, vp, i64, vp, i64, vp, vp, i64, i32, f32, i32, i32, vp, C.c_size_t, vp]),
    "ir_adain_affine_from_partials": (C.c_int, [i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, vp, vp, f32, vp, vp, vp]),
    "ir_token_stats_from_partials": (C.c_int, [i32, i32, i32, vp, i32, vp, vp, vp]),
    "ir_zero_invalid_refs": (C.c_int, [i32, i32, i32, i32, vp, vp, i64, i64, i64, i64,
                                       vp, i64, i64, i64, i64, vp]),
}

_lock = threading.Lock()
_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise ImportError if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with instantrestore_amd/csrc/build.sh "
                "(or __graft_entry__.build()). There is no CPU fallback for this path."
            )
        # torch ships its own libamdhip64; loading it FIRST makes the loader hand the same runtime to this
        # library (same SONAME).  The other order puts two HIP runtimes in one process and the second one
        # finds "no ROCm-capable device".
        import torch  # noqa: F401
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64.so not found
            raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError => stale library, also loud
            fn.restype, fn.argtypes = res, args
        got = handle.ir_abi_version()
        if got != ABI_VERSION:
            raise ImportError(f"{LIB_PATH}: ABI version {got}, expected {ABI_VERSION}; rebuild")
        _lib = handle
    return _lib


class IRError(RuntimeError):
    """a C-ABI call returned a negative status"""


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ir_last_error_string().decode(errors="replace")
        raise IRError(f"{what} failed with status {rc}: {msg}")
