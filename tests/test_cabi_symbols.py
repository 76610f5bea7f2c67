"""The C-ABI library loads (CPU-only box: no compute calls) and exports every symbol that
include/instantrestore_hip.h declares; the ctypes mirror agrees with the header."""
import ctypes as C
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "instantrestore_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ir_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from instantrestore_amd import _lib
    names = _declared()
    assert {"ir_shared_attn_fwd", "ir_adain_stats", "ir_adain_apply", "ir_attn_probs", "ir_token_stats",
            "ir_zero_invalid_refs", "ir_abi_version", "ir_last_error_string"} <= set(names)
    lib = _lib.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in instantrestore_amd/_lib.py"
    assert sorted(_lib.SYMBOLS) == names
    assert lib.ir_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define IR_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert b"gfx950" in lib.ir_build_info()


def test_struct_mirror_matches_header_field_order():
    from instantrestore_amd import _lib
    text = open(HEADER).read()
    body = re.search(r"typedef struct ir_shared_attn_args \{(.*?)\} ir_shared_attn_args;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split()[-1] if "," not in decl else None
        if names is None:
            first, *rest = decl.split(",")
            fields.append(first.split()[-1].lstrip("*"))
            fields += [r.strip().lstrip("*") for r in rest]
        else:
            fields.append(names.lstrip("*"))
    assert fields == [f[0] for f in _lib.SharedAttnArgs._fields_]
    assert C.sizeof(_lib.SharedAttnArgs) == 10 * 4 + 9 * 8 + 20 * 8 + 16 + 8 + 8 + 8   # + valid_refs (ABI v8) + seg_mass (ABI v9)


def test_invalid_arguments_are_rejected_without_a_gpu():
    """validation happens before any launch: callable on the CPU-only box"""
    from instantrestore_amd import _lib
    lib = _lib.lib()
    a = _lib.SharedAttnArgs()
    assert lib.ir_shared_attn_fwd(None, None) == -1
    a.struct_size = 7
    assert lib.ir_shared_attn_fwd(C.byref(a), None) == -1 and b"ABI mismatch" in lib.ir_last_error_string()
    a.struct_size = C.sizeof(a)
    a.dtype = 5
    assert lib.ir_shared_attn_fwd(C.byref(a), None) == -2
    a.dtype, a.batch, a.heads, a.len_q = 1, 1, 1, 8
    assert lib.ir_shared_attn_fwd(C.byref(a), None) == -1          # empty K/V sequence
    assert lib.ir_adain_stats_workspace_bytes(8, 5, 4096, 4, 4096) == 8 * 5 * 5 * 16 * 128 * 4
    with pytest.raises(_lib.IRError):
        _lib.check(-1, "demo")


def test_abi_v8_entry_points_validate_without_a_gpu():
    """round 5 (ABI v8): the dump kernels named by the caller, the per-segment mass and the valid-counts field refuse bad
    arguments before any launch; the automatic tile choice takes the split-K tile exactly where it was measured ahead"""
    from instantrestore_amd import _lib
    lib = _lib.lib()
    a = _lib.SharedAttnArgs()
    a.struct_size = C.sizeof(a)
    a.dtype, a.batch, a.heads, a.len_q, a.len_self, a.flags, a.scale = 1, 1, 1, 64, 64, 1, 0.125
    buf = (C.c_char * 65536)()
    ptr = C.cast(C.byref(buf, 64 - C.addressof(buf) % 64), C.c_void_p)        # a 64-byte aligned host address (never dereferenced)
    a.q = a.k_self = a.v_self = ptr
    a.q_sb = a.ks_sb = a.vs_sb = 64 * 64
    a.q_sl = a.ks_sl = a.vs_sl = 64
    a.q_sh = a.ks_sh = a.vs_sh = 64
    assert lib.ir_attn_probs_ex(C.byref(a), None, 0, None) == -1 and b"probs/lse" in lib.ir_last_error_string()
    a.lse = ptr
    assert lib.ir_attn_probs_ex(C.byref(a), ptr, 7, None) == -2                                # unknown kernel id
    assert lib.ir_attn_probs_ex(C.byref(a), ptr, -1, None) == -2
    a.len_self = 60                                                                             # 60 keys: rows of P not 16-byte aligned
    assert lib.ir_attn_probs_ex(C.byref(a), ptr, 2, None) == -2 and b"multiples of 8" in lib.ir_last_error_string()
    assert lib.ir_attn_segment_mass(C.byref(a), None, None) == -1
    a.len_self = 64
    a.n_refs, a.len_ref = 2, 64
    a.k_ref = a.v_ref = ptr
    a.kr_sb = a.vr_sb = 2 * 64 * 64
    a.kr_sn = a.vr_sn = 64 * 64
    a.kr_sl = a.vr_sl = 64
    a.kr_sh = a.vr_sh = 64
    a.out = ptr
    a.o_sb, a.o_sl, a.o_sh = 64 * 64, 64, 64
    a.valid_refs = C.cast(C.c_void_p(ptr.value + 2), C.c_void_p)                                # not 4-byte aligned
    assert lib.ir_shared_attn_fwd(C.byref(a), None) == -2 and b"valid_refs" in lib.ir_last_error_string()
    a.valid_refs = None
    a.seg_mass = C.cast(C.c_void_p(ptr.value + 2), C.c_void_p)                                  # ABI v9: not 4-byte aligned
    assert lib.ir_shared_attn_fwd(C.byref(a), None) == -2 and b"seg_mass" in lib.ir_last_error_string()
    a.seg_mass = None
    # tile picker: M = 2048 x N = 1280 x K = 1280 (160 tiles of 128 x 128 on 256 CUs) takes the split-K tile (id 9); the same
    # rows at N = 3840 (480 tiles) and the K = 640 shapes do not
    assert lib.ir_linear_kernel_for(2048, 1280, 1280, 1) == 9
    assert lib.ir_linear_kernel_for(2048, 3840, 1280, 0) == 3 and lib.ir_linear_kernel_for(8192, 640, 640, 1) == 3
    assert lib.ir_linear_kernel_for(8192, 3840, 1280, 0) == 8


def test_statistics_tail_entry_points_validate_without_a_gpu():
    """ABI 6 (round 4): the q/k/v projection that leaves the AdaIN token statistics behind and the merges of its partials -
    shape rules answered and bad arguments rejected before any launch"""
    from instantrestore_amd import _lib
    lib = _lib.lib()
    # every projection kernel gives a wave 64 rows of Y: the statistics block; M must be whole blocks, N whole heads
    assert lib.ir_linear_stats_rows(131072, 960, 320, 0) == 64 and lib.ir_linear_stats_rows(8192, 3840, 1280, 0) == 64
    assert lib.ir_linear_stats_rows(2048, 3840, 1280, 0) == 64 and lib.ir_linear_stats_rows(100, 3840, 1280, 0) == 0
    assert lib.ir_linear_stats_rows(8192, 96, 320, 0) == 0                       # N % 64 != 0
    buf = (C.c_float * 16)()
    ptr = C.cast(buf, C.c_void_p)
    assert lib.ir_linear_fwd_stats(1, 0, 8192, 3840, 1280, ptr, 1280, ptr, 1280, None, ptr, 3840, 0, 1.0, 2560, 1280, None, 0, None) == -1
    assert b"stats_ws" in lib.ir_last_error_string()
    assert lib.ir_adain_affine_from_partials(8, 5, 4, 4096, 4096, None, 64, None, 64, None, None, None, 1e-5, ptr, ptr, None) == -1
    assert lib.ir_adain_affine_from_partials(8, 5, 4, 4096, 4096, ptr, 60, ptr, 64, None, None, None, 1e-5, ptr, ptr, None) == -1   # 4096 % 60
    assert lib.ir_adain_affine_from_partials(8, 5, 4, 4096, 4096, ptr, 64, None, 0, None, None, None, 1e-5, ptr, ptr, None) == -1   # no content statistics
    assert lib.ir_adain_affine_from_partials(1, 5, 4, 65536, 65536, ptr, 64, ptr, 64, None, None, None, 1e-5, ptr, ptr, None) == -2  # > 256 partials per matrix
    assert lib.ir_token_stats_from_partials(8, 5, 4096, None, 64, ptr, ptr, None) == -1
    assert lib.ir_token_stats_from_partials(8, 5, 100, ptr, 64, ptr, ptr, None) == -1


def test_variant_env_var_is_applied_at_load():
    """IR_ATTN_VARIANT=<n> selects a kernel variant for the whole process without code changes (the value rides
    in the per-call `tuning` field; the C library keeps no such state)"""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from instantrestore_amd import ops; "
            "print(ops.set_attn_variant(0))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, IR_ATTN_VARIANT="11")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == "11"


def test_mfma_stream_hook_validates_without_a_gpu():
    """ir_bench_mfma_stream (ABI v7, bench.py `roofline.at_power_cap`) refuses bad arguments before it touches the device"""
    from instantrestore_amd import _lib
    lib = _lib.lib()
    out = C.c_float(0.0)
    need = int(lib.ir_bench_mfma_stream_scratch_bytes())
    assert need >= 256 * 512 * 4
    fake = C.c_void_p(1 << 20)      # never dereferenced: every call below fails validation
    assert lib.ir_bench_mfma_stream(1, 0, 0, 1, fake, need, None, C.byref(out)) == -1        # IR_ERR_INVALID_ARG: iters
    assert lib.ir_bench_mfma_stream(1, 0, 10, 0, fake, need, None, C.byref(out)) == -1       # launches
    assert lib.ir_bench_mfma_stream(1, 0, 10, 1, None, need, None, C.byref(out)) == -1       # scratch
    assert lib.ir_bench_mfma_stream(7, 0, 10, 1, fake, need, None, C.byref(out)) == -2       # IR_ERR_UNSUPPORTED: dtype
    assert lib.ir_bench_mfma_stream(1, 0, 10, 1, fake, 16, None, C.byref(out)) == -4         # IR_ERR_WORKSPACE
