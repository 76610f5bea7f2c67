// c_abi.hip - the extern "C" surface declared in include/instantrestore_hip.h.
// Argument validation, parameter-block construction, launches. No exceptions, no global state
// other than a thread-local error string.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/instantrestore_hip.h"
#include "ir_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
bool stride_ok(int64_t s) { return s >= 0 && (s % 8) == 0; }  // keeps every head row 16-B aligned

int build_attn_params(const ir_shared_attn_args* a, AttnKParams* p, bool need_out) {
  if (a == nullptr) return fail(IR_ERR_INVALID_ARG, "args is NULL");
  if (a->struct_size != sizeof(ir_shared_attn_args))
    return fail(IR_ERR_INVALID_ARG, "struct_size %u != %zu (ABI mismatch)", a->struct_size, sizeof(ir_shared_attn_args));
  if (a->dtype != IR_DTYPE_F16 && a->dtype != IR_DTYPE_BF16)
    return fail(IR_ERR_UNSUPPORTED, "dtype %d: only fp16 (0) and bf16 (1) are implemented", a->dtype);
  if (a->batch <= 0 || a->heads <= 0 || a->len_q <= 0) return fail(IR_ERR_INVALID_ARG, "batch/heads/len_q must be > 0");
  if (a->n_refs < 0 || a->len_self < 0 || a->len_ref < 0) return fail(IR_ERR_INVALID_ARG, "negative length");
  const bool inc = (a->flags & IR_FLAG_INCLUDE_SELF) != 0;
  if ((a->flags & ~(IR_FLAG_INCLUDE_SELF | IR_FLAG_Q_PRESCALED | IR_FLAG_OUT_F32)) != 0) return fail(IR_ERR_INVALID_ARG, "unknown flag bits 0x%x", a->flags);
  if (a->reserved != 0) return fail(IR_ERR_INVALID_ARG, "reserved must be 0");
  if (!ir_attn_variant_available(a->tuning)) return fail(IR_ERR_UNSUPPORTED, "tuning value %d is not available in this build", a->tuning);
  if (inc && a->len_self <= 0) return fail(IR_ERR_INVALID_ARG, "INCLUDE_SELF with len_self == 0");
  if (a->n_refs > 0 && a->len_ref <= 0) return fail(IR_ERR_INVALID_ARG, "n_refs > 0 with len_ref == 0");
  if (!inc && a->n_refs == 0) return fail(IR_ERR_INVALID_ARG, "empty key/value sequence");
  if (a->q == nullptr || (need_out && a->out == nullptr)) return fail(IR_ERR_INVALID_ARG, "q/out is NULL");
  if (inc && (a->k_self == nullptr || a->v_self == nullptr)) return fail(IR_ERR_INVALID_ARG, "k_self/v_self is NULL");
  if (a->n_refs > 0 && (a->k_ref == nullptr || a->v_ref == nullptr)) return fail(IR_ERR_INVALID_ARG, "k_ref/v_ref is NULL");
  if ((a->adain_a == nullptr) != (a->adain_b == nullptr)) return fail(IR_ERR_INVALID_ARG, "adain_a and adain_b must both be set or both be NULL");
  if (a->adain_a != nullptr && a->n_refs == 0) return fail(IR_ERR_INVALID_ARG, "AdaIN affine without references");
  if (a->valid_refs != nullptr && (reinterpret_cast<uintptr_t>(a->valid_refs) & 3u) != 0) return fail(IR_ERR_UNSUPPORTED, "valid_refs must be 4-byte aligned");
  if (a->seg_mass != nullptr && (reinterpret_cast<uintptr_t>(a->seg_mass) & 3u) != 0) return fail(IR_ERR_UNSUPPORTED, "seg_mass must be 4-byte aligned");
  const void* ptrs[] = {a->q, a->k_self, a->v_self, a->k_ref, a->v_ref, a->out, a->adain_a, a->adain_b};
  for (const void* q : ptrs)
    if (q != nullptr && !aligned16(q)) return fail(IR_ERR_UNSUPPORTED, "pointer %p is not 16-byte aligned", q);
  const int64_t strides[] = {a->q_sb, a->q_sl, a->q_sh, a->ks_sb, a->ks_sl, a->ks_sh, a->vs_sb, a->vs_sl, a->vs_sh,
                             a->kr_sb, a->kr_sn, a->kr_sl, a->kr_sh, a->vr_sb, a->vr_sn, a->vr_sl, a->vr_sh,
                             a->o_sb, a->o_sl, a->o_sh};
  for (int64_t s : strides)
    if (!stride_ok(s)) return fail(IR_ERR_UNSUPPORTED, "stride %lld must be a non-negative multiple of 8 elements", (long long)s);

  memset(p, 0, sizeof(*p));
  p->q = a->q; p->k_self = a->k_self; p->v_self = a->v_self; p->k_ref = a->k_ref; p->v_ref = a->v_ref;
  p->aa = a->adain_a; p->ab = a->adain_b; p->out = a->out; p->lse = a->lse;
  p->valid = a->n_refs > 0 ? a->valid_refs : nullptr;
  p->seg_cum = need_out ? a->seg_mass : nullptr;   // a by-product of the forward launch only (ir_attn_probs / _segment_mass ignore it)
  p->nseg_out = (inc ? 1 : 0) + a->n_refs;
  p->q_sb = a->q_sb; p->q_sl = a->q_sl; p->q_sh = a->q_sh;
  p->ks_sb = a->ks_sb; p->ks_sl = a->ks_sl; p->ks_sh = a->ks_sh;
  p->vs_sb = a->vs_sb; p->vs_sl = a->vs_sl; p->vs_sh = a->vs_sh;
  p->kr_sb = a->kr_sb; p->kr_sn = a->kr_sn; p->kr_sl = a->kr_sl; p->kr_sh = a->kr_sh;
  p->vr_sb = a->vr_sb; p->vr_sn = a->vr_sn; p->vr_sl = a->vr_sl; p->vr_sh = a->vr_sh;
  p->o_sb = a->o_sb; p->o_sl = a->o_sl; p->o_sh = a->o_sh;
  p->B = a->batch; p->H = a->heads; p->Lq = a->len_q; p->Ls = a->len_self; p->N = a->n_refs; p->Lr = a->len_ref;
  p->include_self = inc ? 1 : 0;
  p->q_prescaled = (a->flags & IR_FLAG_Q_PRESCALED) ? 1 : 0;
  p->out_f32 = (a->flags & IR_FLAG_OUT_F32) ? 1 : 0;
  if (p->q_prescaled && a->tuning != IR_TUNE_DEFAULT && a->tuning != IR_TUNE_W64X8 && a->tuning != IR_TUNE_W128 && a->tuning != IR_TUNE_PIPE32_PRESCALE_Q &&
      a->tuning != IR_TUNE_PIPE32_POSTCHECK && !(a->tuning >= IR_TUNE_W64_ABL_FIRST && a->tuning < IR_TUNE_W64_ABL_FIRST + 16))
    return fail(IR_ERR_UNSUPPORTED, "IR_FLAG_Q_PRESCALED is implemented by the W128, W64X8, PIPE32_PRESCALE_Q and PIPE32_POSTCHECK kernels only");
  if (a->tuning == IR_TUNE_W128 && !p->q_prescaled)
    return fail(IR_ERR_UNSUPPORTED, "IR_TUNE_W128 needs IR_FLAG_Q_PRESCALED");
  if (!p->q_prescaled && a->tuning == IR_TUNE_PIPE32_POSTCHECK)
    return fail(IR_ERR_UNSUPPORTED, "IR_TUNE_PIPE32_POSTCHECK needs IR_FLAG_Q_PRESCALED (its scores must already carry the reference)");
  p->tiles_self = inc ? (a->len_self + IR_KV_TILE - 1) / IR_KV_TILE : 0;
  p->tiles_ref = a->n_refs > 0 ? (a->len_ref + IR_KV_TILE - 1) / IR_KV_TILE : 0;
  p->ntiles = p->tiles_self + a->n_refs * p->tiles_ref;
  p->lkv = (inc ? a->len_self : 0) + a->n_refs * a->len_ref;
  p->scale = a->scale;
  p->scale_log2 = a->scale * 1.4426950408889634f;
  if (a->workspace != nullptr && !aligned16(a->workspace)) return fail(IR_ERR_UNSUPPORTED, "workspace must be 16-byte aligned");
  p->ws = (float*)a->workspace;
  p->ws_bytes = a->workspace != nullptr ? (size_t)a->workspace_bytes : 0;
  if (a->tuning == IR_TUNE_W128 && !ir_attn_w128_supports(*p))
    return fail(IR_ERR_UNSUPPORTED, "IR_TUNE_W128 takes segment lengths that are multiples of 64 keys, no valid_refs and no seg_mass");
  const int64_t blocks = (int64_t)a->batch * a->heads * ((a->len_q + 127) / 128);
  if (blocks > 0x7fffffffLL) return fail(IR_ERR_UNSUPPORTED, "grid too large");
  return IR_OK;
}

}  // namespace

extern "C" {

int ir_abi_version(void) { return IR_ABI_VERSION; }

const char* ir_build_info(void) { return "instantrestore_hip gfx950 (CDNA4) hipcc " __VERSION__ " built " __DATE__; }

const char* ir_last_error_string(void) { return g_err; }

// 8 XCDs x 64 slots pieces of up to 256 rows, 64 fp32 of O + (max, sum) per row
const char* ir_shared_attn_kernel_name(const ir_shared_attn_args* args) {
  AttnKParams p;
  if (build_attn_params(args, &p, false) != IR_OK) return "";
  const int v = args->tuning & 31;
  const bool fold = p.aa != nullptr;
  const bool w64 = (v == 0 && ir_attn_default_is_w64(p)) || v == 13;
  if ((v == 16 || (v == 0 && ir_attn_default_is_w128(p))) && ir_attn_w128_supports(p))
    return fold ? "shared_attn_fwd_w128_kernel<128 rows/wave, one wave per SIMD, hand-placed stream, pre-scaled Q, AdaIN ratio-frame fold>"
                : "shared_attn_fwd_w128_kernel<128 rows/wave, one wave per SIMD, hand-placed stream, pre-scaled Q>";
  if (p.q_prescaled) {   // the dispatch of ir_launch_shared_attn_fwd, restated for reporting
    if (w64) return fold ? "shared_attn_fwd_w64_kernel<64 rows/wave, 8 waves, pre-scaled Q (reference through the MFMA C operand, checked after the exponentials), AdaIN ratio-frame fold>"
                         : "shared_attn_fwd_w64_kernel<64 rows/wave, 8 waves, pre-scaled Q (reference through the MFMA C operand, checked after the exponentials)>";
    if (v == 11 || v == 0) return fold ? "shared_attn_fwd_pipe_kernel<4 waves, lazy max, pre-scaled Q, AdaIN fold>" : "shared_attn_fwd_pipe_kernel<4 waves, lazy max, pre-scaled Q>";
    if (v == 18) return fold ? "shared_attn_fwd_pipe_kernel<4 waves, pre-scaled Q, reference checked after the exponentials, AdaIN fold>"
                                       : "shared_attn_fwd_pipe_kernel<4 waves, pre-scaled Q, reference checked after the exponentials>";
  }
  switch (v) {
    case 0: return ir_attn_default_is_w64(p) ? (fold ? "shared_attn_fwd_w64_kernel<64 rows/wave, 8 waves, AdaIN ratio-frame fold>" : "shared_attn_fwd_w64_kernel<64 rows/wave, 8 waves>")
                                             : (fold ? "shared_attn_fwd_pipe_kernel<4 waves, lazy max, early QK, AdaIN fold>" : "shared_attn_fwd_pipe_kernel<4 waves, lazy max, early QK>");
    case 12: return fold ? "shared_attn_fwd_w64_kernel<64 rows/wave, 4 waves, AdaIN ratio-frame fold>" : "shared_attn_fwd_w64_kernel<64 rows/wave, 4 waves>";
    case 13: return fold ? "shared_attn_fwd_w64_kernel<64 rows/wave, 8 waves, AdaIN ratio-frame fold>" : "shared_attn_fwd_w64_kernel<64 rows/wave, 8 waves>";
    case 11: return "shared_attn_fwd_pipe_kernel<4 waves, lazy max, pre-scaled Q (Q rounded in the kernel)>";
    case 14: return fold ? "shared_attn_fwd_pipe_kernel<4 waves, lazy max, early QK, AdaIN fold>" : "shared_attn_fwd_pipe_kernel<4 waves, lazy max, early QK>";
    case 10: return "shared_attn_fwd_pipe_kernel<4 waves, lazy max>";
    case 7: return "shared_attn_fwd_pipe_kernel<4 waves, exact rescale>";
    default: return "shared_attn_fwd (tuning variant)";
  }
}

// 8 XCDs x 64 pieces x 512 rows x (64 + 2) floats: the largest remainder split of any kernel (512-row items)
size_t ir_shared_attn_workspace_bytes(void) { return (size_t)8 * 64 * 512 * 66 * sizeof(float); }

int ir_shared_attn_fwd(const ir_shared_attn_args* args, void* stream) {
  AttnKParams p;
  const int rc = build_attn_params(args, &p, true);
  if (rc != IR_OK) return rc;
  const hipError_t e = ir_launch_shared_attn_fwd(p, args->dtype, args->tuning, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "shared_attn_fwd launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_time_shared_attn_fwd(const ir_shared_attn_args* args, int32_t iters, void* stream, float* ms_per_launch) {
  if (ms_per_launch == nullptr || iters <= 0) return fail(IR_ERR_INVALID_ARG, "iters/ms_per_launch");
  AttnKParams p;
  const int rc = build_attn_params(args, &p, true);
  if (rc != IR_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(IR_ERR_LAUNCH, "hipEventCreate failed");
  hipError_t e = hipEventRecord(e0, s);
  for (int i = 0; i < iters && e == hipSuccess; ++i) e = ir_launch_shared_attn_fwd(p, args->dtype, args->tuning, s);
  if (e == hipSuccess) e = hipEventRecord(e1, s);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "timed launch: %s", hipGetErrorString(e));
  *ms_per_launch = ms / (float)iters;
  return IR_OK;
}

static int bench_blocks() {   // one 8-wave workgroup per CU = two waves per SIMD
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return cus;
}

size_t ir_bench_mfma_stream_scratch_bytes(void) { return (size_t)1024 * 512 * sizeof(float); }   // up to 1024 CUs

int ir_bench_mfma_stream(int32_t dtype, int32_t zero_operands, int32_t iters, int32_t launches, void* scratch, size_t scratch_bytes,
                         void* stream, float* tflops) {
  if (dtype != IR_DTYPE_F16 && dtype != IR_DTYPE_BF16) return fail(IR_ERR_UNSUPPORTED, "dtype %d: fp16 (0) / bf16 (1)", dtype);
  if (iters <= 0 || launches <= 0 || tflops == nullptr || scratch == nullptr) return fail(IR_ERR_INVALID_ARG, "iters/launches/scratch/tflops");
  const int blocks = bench_blocks();
  if (blocks > 1024 || scratch_bytes < (size_t)blocks * 512 * sizeof(float)) return fail(IR_ERR_WORKSPACE, "scratch %zu < %zu bytes", scratch_bytes, (size_t)blocks * 512 * sizeof(float));
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail(IR_ERR_LAUNCH, "hipEventCreate failed");
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = ir_launch_bench_mfma_stream(dtype, zero_operands, iters, blocks, (float*)scratch, s);   // untimed: clocks settle
  if (e == hipSuccess) e = hipEventRecord(e0, s);
  for (int i = 0; i < launches && e == hipSuccess; ++i) e = ir_launch_bench_mfma_stream(dtype, zero_operands, iters, blocks, (float*)scratch, s);
  if (e == hipSuccess) e = hipEventRecord(e1, s);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "mfma stream: %s", hipGetErrorString(e));
  const double flops = (double)launches * iters * 16.0 * 32768.0 * (double)blocks * 8.0;   // 2 * 32 * 32 * 16 per MFMA, 8 waves per workgroup
  *tflops = (float)(flops / ((double)ms * 1e-3) / 1e12);
  return IR_OK;
}

int ir_attn_probs_ex(const ir_shared_attn_args* args, void* probs, int32_t kernel, void* stream) {
  AttnKParams p;
  const int rc = build_attn_params(args, &p, false);
  if (rc != IR_OK) return rc;
  if (probs == nullptr || args->lse == nullptr) return fail(IR_ERR_INVALID_ARG, "probs/lse is NULL");
  if (kernel < IR_PROBS_AUTO || kernel > IR_PROBS_LINES32_K256) return fail(IR_ERR_UNSUPPORTED, "attn_probs kernel %d", kernel);
  p.probs = probs;
  if (p.q_prescaled) p.scale_log2 = 1.0f;   // IR_FLAG_Q_PRESCALED: the products of q and k ARE the exponents (`scale` is the LSE's unit only)
  if (kernel >= IR_PROBS_LINES64 && !ir_attn_probs_uses_lines(p))
    return fail(IR_ERR_UNSUPPORTED, "the line kernel needs len_self, len_ref (and so Lkv) to be multiples of 8 and probs 16-byte aligned");
  const hipError_t e = ir_launch_attn_probs(p, args->dtype, kernel, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "attn_probs launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_attn_probs(const ir_shared_attn_args* args, void* probs, void* stream) { return ir_attn_probs_ex(args, probs, IR_PROBS_AUTO, stream); }

int ir_attn_segment_mass(const ir_shared_attn_args* args, float* mass, void* stream) {
  AttnKParams p;
  const int rc = build_attn_params(args, &p, false);
  if (rc != IR_OK) return rc;
  if (mass == nullptr || args->lse == nullptr) return fail(IR_ERR_INVALID_ARG, "mass/lse is NULL");
  if (p.q_prescaled) p.scale_log2 = 1.0f;
  const hipError_t e = ir_launch_attn_segment_mass(p, args->dtype, mass, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "attn_segment_mass launch: %s", hipGetErrorString(e));
  return IR_OK;
}

static int adain_nchunk(int32_t len_self, int32_t len_ref) {
  const int a = (len_self + IR_ADAIN_ROWS - 1) / IR_ADAIN_ROWS;
  const int b = (len_ref + IR_ADAIN_ROWS - 1) / IR_ADAIN_ROWS;
  const int m = a > b ? a : b;
  return m > 0 ? m : 1;
}

size_t ir_adain_stats_workspace_bytes(int32_t batch, int32_t heads, int32_t len_self, int32_t n_refs, int32_t len_ref) {
  if (batch <= 0 || heads <= 0 || n_refs < 0) return 0;
  return (size_t)batch * (size_t)(1 + n_refs) * (size_t)heads * (size_t)adain_nchunk(len_self, len_ref) * 128u * sizeof(float);
}

int ir_adain_stats(int32_t dtype, int32_t batch, int32_t heads, int32_t len_self, int32_t n_refs, int32_t len_ref,
                   const void* v_self, int64_t vs_sb, int64_t vs_sl, int64_t vs_sh,
                   const void* v_ref, int64_t vr_sb, int64_t vr_sn, int64_t vr_sl, int64_t vr_sh,
                   float eps, float* a, float* b, void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != IR_DTYPE_F16 && dtype != IR_DTYPE_BF16) return fail(IR_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (batch <= 0 || heads <= 0 || len_self <= 0 || n_refs <= 0 || len_ref <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!v_self || !v_ref || !a || !b || !workspace) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (!aligned16(v_self) || !aligned16(v_ref)) return fail(IR_ERR_UNSUPPORTED, "v_self/v_ref must be 16-byte aligned");
  const int64_t st[] = {vs_sb, vs_sl, vs_sh, vr_sb, vr_sn, vr_sl, vr_sh};
  for (int64_t s : st) if (!stride_ok(s)) return fail(IR_ERR_UNSUPPORTED, "stride %lld must be a non-negative multiple of 8", (long long)s);
  const size_t need = ir_adain_stats_workspace_bytes(batch, heads, len_self, n_refs, len_ref);
  if (workspace_bytes < need) return fail(IR_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  AdainKParams p;
  memset(&p, 0, sizeof(p));
  p.v_self = v_self; p.v_ref = v_ref;
  p.vs_sb = vs_sb; p.vs_sl = vs_sl; p.vs_sh = vs_sh;
  p.vr_sb = vr_sb; p.vr_sn = vr_sn; p.vr_sl = vr_sl; p.vr_sh = vr_sh;
  p.ws = (float*)workspace; p.a = a; p.b = b;
  p.B = batch; p.H = heads; p.Ls = len_self; p.N = n_refs; p.Lr = len_ref;
  p.nchunk = adain_nchunk(len_self, len_ref);
  p.eps = eps;
  const hipError_t e = ir_launch_adain_stats(p, dtype, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "adain_stats launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_adain_stats_cached(int32_t dtype, int32_t batch, int32_t heads, int32_t len_self, int32_t n_refs,
                          const void* v_self, int64_t vs_sb, int64_t vs_sl, int64_t vs_sh,
                          const float* content_mean, const float* content_std,
                          float eps, float* a, float* b, void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != IR_DTYPE_F16 && dtype != IR_DTYPE_BF16) return fail(IR_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (batch <= 0 || heads <= 0 || len_self <= 0 || n_refs <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!v_self || !content_mean || !content_std || !a || !b || !workspace) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (!aligned16(v_self)) return fail(IR_ERR_UNSUPPORTED, "v_self must be 16-byte aligned");
  const int64_t st[] = {vs_sb, vs_sl, vs_sh};
  for (int64_t s : st) if (!stride_ok(s)) return fail(IR_ERR_UNSUPPORTED, "stride %lld must be a non-negative multiple of 8", (long long)s);
  const size_t need = ir_adain_stats_workspace_bytes(batch, heads, len_self, 0, len_self);
  if (workspace_bytes < need) return fail(IR_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  AdainKParams p;
  memset(&p, 0, sizeof(p));
  p.v_self = v_self; p.v_ref = v_self;
  p.vs_sb = vs_sb; p.vs_sl = vs_sl; p.vs_sh = vs_sh;
  p.ws = (float*)workspace; p.a = a; p.b = b;
  p.B = batch; p.H = heads; p.Ls = len_self; p.N = n_refs; p.Lr = len_self;
  p.nchunk = adain_nchunk(len_self, len_self);
  p.eps = eps;
  p.cmean = content_mean; p.cstd = content_std;
  const hipError_t e = ir_launch_adain_stats_cached(p, dtype, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "adain_stats_cached launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_token_stats(int32_t dtype, int32_t batch, int32_t heads, int32_t n_mats, int32_t len,
                   const void* x, int64_t x_sb, int64_t x_sn, int64_t x_sl, int64_t x_sh,
                   float* mean, float* std, void* workspace, size_t workspace_bytes, void* stream) {
  if (dtype != IR_DTYPE_F16 && dtype != IR_DTYPE_BF16) return fail(IR_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (batch <= 0 || heads <= 0 || n_mats <= 0 || len <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!x || !mean || !std || !workspace) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (!aligned16(x)) return fail(IR_ERR_UNSUPPORTED, "x must be 16-byte aligned");
  const int64_t st[] = {x_sb, x_sn, x_sl, x_sh};
  for (int64_t s : st) if (!stride_ok(s)) return fail(IR_ERR_UNSUPPORTED, "stride %lld must be a non-negative multiple of 8", (long long)s);
  const size_t need = ir_adain_stats_workspace_bytes(batch, heads, len, n_mats - 1, len);
  if (workspace_bytes < need) return fail(IR_ERR_WORKSPACE, "workspace %zu < %zu bytes", workspace_bytes, need);
  // matrix 0 of every batch entry plays the "self" role of the partial kernel, 1.. the "refs"
  AdainKParams p;
  memset(&p, 0, sizeof(p));
  p.v_self = x; p.v_ref = (const char*)x + x_sn * 2;
  p.vs_sb = x_sb; p.vs_sl = x_sl; p.vs_sh = x_sh;
  p.vr_sb = x_sb; p.vr_sn = x_sn; p.vr_sl = x_sl; p.vr_sh = x_sh;
  p.ws = (float*)workspace; p.a = mean; p.b = std;
  p.B = batch; p.H = heads; p.Ls = len; p.N = n_mats - 1; p.Lr = len;
  p.nchunk = adain_nchunk(len, len);
  const hipError_t e = ir_launch_token_stats(p, dtype, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "token_stats launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_adain_apply(int32_t dtype, int32_t batch, int32_t heads, int32_t n_refs, int32_t len,
                   const void* x, int64_t x_sb, int64_t x_sn, int64_t x_sl, int64_t x_sh,
                   const float* a, const float* b,
                   void* y, int64_t y_sb, int64_t y_sn, int64_t y_sl, int64_t y_sh, void* stream) {
  if (dtype != IR_DTYPE_F16 && dtype != IR_DTYPE_BF16) return fail(IR_ERR_UNSUPPORTED, "dtype %d", dtype);
  if (batch <= 0 || heads <= 0 || n_refs <= 0 || len <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!x || !y || !a || !b) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (!aligned16(x) || !aligned16(y)) return fail(IR_ERR_UNSUPPORTED, "x/y must be 16-byte aligned");
  const int64_t st[] = {x_sb, x_sn, x_sl, x_sh, y_sb, y_sn, y_sl, y_sh};
  for (int64_t s : st) if (!stride_ok(s)) return fail(IR_ERR_UNSUPPORTED, "stride %lld must be a non-negative multiple of 8", (long long)s);
  AdainApplyKParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.y = y; p.a = a; p.b = b;
  p.x_sb = x_sb; p.x_sn = x_sn; p.x_sl = x_sl; p.x_sh = x_sh;
  p.y_sb = y_sb; p.y_sn = y_sn; p.y_sl = y_sl; p.y_sh = y_sh;
  p.B = batch; p.H = heads; p.N = n_refs; p.L = len;
  const hipError_t e = ir_launch_adain_apply(p, dtype, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "adain_apply launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_zero_invalid_refs(int32_t batch, int32_t heads, int32_t n_refs, int32_t len, const int32_t* valid,
                         void* k, int64_t k_sb, int64_t k_sn, int64_t k_sl, int64_t k_sh,
                         void* v, int64_t v_sb, int64_t v_sn, int64_t v_sl, int64_t v_sh, void* stream) {
  if (batch <= 0 || heads <= 0 || n_refs <= 0 || len <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!valid || !k || !v) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (!aligned16(k) || !aligned16(v)) return fail(IR_ERR_UNSUPPORTED, "k/v must be 16-byte aligned");
  const int64_t st[] = {k_sb, k_sn, k_sl, k_sh, v_sb, v_sn, v_sl, v_sh};
  for (int64_t s : st) if (!stride_ok(s)) return fail(IR_ERR_UNSUPPORTED, "stride %lld must be a non-negative multiple of 8", (long long)s);
  ZeroRefsKParams p;
  memset(&p, 0, sizeof(p));
  p.k = k; p.v = v; p.valid = valid;
  p.k_sb = k_sb; p.k_sn = k_sn; p.k_sl = k_sl; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sl = v_sl; p.v_sh = v_sh;
  p.B = batch; p.H = heads; p.N = n_refs; p.L = len;
  const hipError_t e = ir_launch_zero_refs(p, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "zero_refs launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_tensor2im_u8(int32_t dtype, int32_t batch, int32_t channels, int32_t height, int32_t width, const void* x,
                    int64_t x_sb, int64_t x_sc, int64_t x_sh, int64_t x_sw, void* out_u8, void* stream) {
  if (dtype < 0 || dtype > 2) return fail(IR_ERR_UNSUPPORTED, "dtype %d (0 f16, 1 bf16, 2 f32)", dtype);
  if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!x || !out_u8) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  const hipError_t e = ir_launch_tensor2im(x, out_u8, dtype, x_sb, x_sc, x_sh, x_sw, batch, channels, height, width,
                                           (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "tensor2im launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_lanczos_ksize(int32_t in_size, int32_t out_size) {
  if (in_size <= 0 || out_size <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  return ir_host_lanczos_ksize(in_size, out_size);
}

int ir_lanczos_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk) {
  if (in_size <= 0 || out_size <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!bounds || !kk) return fail(IR_ERR_INVALID_ARG, "NULL table");
  ir_host_lanczos_coeffs(in_size, out_size, bounds, kk);
  return IR_OK;
}

int ir_preprocess_lanczos_u8(const ir_image_desc* images, int32_t n_images, int32_t size, int32_t out_dtype,
                             void* out, void* stream) {
  if (out_dtype < 0 || out_dtype > 2) return fail(IR_ERR_UNSUPPORTED, "out_dtype %d (0 f16, 1 bf16, 2 f32)", out_dtype);
  if (!images || !out || n_images <= 0 || size <= 0) return fail(IR_ERR_INVALID_ARG, "NULL pointer or empty batch");
  for (int i = 0; i < n_images; ++i) {
    const ir_image_desc& d = images[i];
    if (!d.src || !d.bounds_h || !d.kk_h || !d.bounds_v || !d.kk_v || !d.tmp)
      return fail(IR_ERR_INVALID_ARG, "image %d: NULL pointer", i);
    if (d.in_h <= 0 || d.in_w <= 0 || d.src_row_bytes < (int64_t)d.in_w * 3)
      return fail(IR_ERR_INVALID_ARG, "image %d: bad source geometry", i);
    if (d.out_h < size || d.out_w < size || d.crop_top < 0 || d.crop_left < 0 || d.crop_top + size > d.out_h ||
        d.crop_left + size > d.out_w)
      return fail(IR_ERR_INVALID_ARG, "image %d: crop %dx%d at (%d,%d) outside the %dx%d resized image", i, size, size,
                  d.crop_top, d.crop_left, d.out_h, d.out_w);
    if (d.ksize_h <= 0 || d.ksize_v <= 0 || d.row_first < 0 || d.row_count <= 0 || d.row_first + d.row_count > d.in_h)
      return fail(IR_ERR_INVALID_ARG, "image %d: bad tap tables / row range", i);
    if (d.col_first < 0 || d.col_count <= 0 || d.col_first + d.col_count > d.in_w)
      return fail(IR_ERR_INVALID_ARG, "image %d: bad column range", i);
    if ((int64_t)d.col_count * 3 > 60000) return fail(IR_ERR_UNSUPPORTED, "image %d: more than 20000 source columns under the crop", i);
    if (reinterpret_cast<uintptr_t>(d.tmp) & 3u) return fail(IR_ERR_UNSUPPORTED, "image %d: tmp must be 4-byte aligned", i);
  }
  for (int first = 0; first < n_images; first += kPreprocessImagesPerLaunch) {
    PreprocessKParams p;
    memset(&p, 0, sizeof(p));
    p.n = n_images - first < kPreprocessImagesPerLaunch ? n_images - first : kPreprocessImagesPerLaunch;
    p.size = size;
    p.first_image = first;
    p.tmp_pitch = (size * 3 + 3) & ~3;
    int max_rows = 0, max_span = 0, max_ksv = 0;
    for (int i = 0; i < p.n; ++i) {
      const ir_image_desc& d = images[first + i];
      ResampleImageK& k = p.img[i];
      k.src = (const unsigned char*)d.src; k.src_row_bytes = d.src_row_bytes;
      k.bounds_h = d.bounds_h; k.kk_h = d.kk_h; k.bounds_v = d.bounds_v; k.kk_v = d.kk_v;
      k.tmp = (unsigned char*)d.tmp; k.ksize_h = d.ksize_h; k.ksize_v = d.ksize_v;
      k.crop_top = d.crop_top; k.crop_left = d.crop_left; k.row_first = d.row_first; k.row_count = d.row_count;
      k.col_first = d.col_first; k.col_count = d.col_count; k.in_h = d.in_h; k.in_w = d.in_w; k.out_w = d.out_w;
      if (d.row_count > max_rows) max_rows = d.row_count;
      if (d.col_count * 3 > max_span) max_span = d.col_count * 3;
      if (d.ksize_v > max_ksv) max_ksv = d.ksize_v;
    }
    const hipError_t e = ir_launch_preprocess(p, max_rows, max_span, max_ksv, out_dtype, out, (hipStream_t)stream);
    if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "preprocess launch: %s", hipGetErrorString(e));
  }
  return IR_OK;
}

int ir_freeu_fourier_filter(int32_t dtype, int64_t planes, int32_t height, int32_t width, const void* x,
                            int64_t x_plane_stride, void* out, int64_t out_plane_stride, int32_t threshold,
                            float scale, void* stream) {
  if (dtype < 0 || dtype > 2) return fail(IR_ERR_UNSUPPORTED, "dtype %d (0 f16, 1 bf16, 2 f32)", dtype);
  if (!x || !out) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (planes <= 0 || height <= 0 || width <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if ((int64_t)height * width > 4096) return fail(IR_ERR_UNSUPPORTED, "plane %dx%d: at most 4096 elements", height, width);
  if (threshold < 1 || 2 * threshold > height || 2 * threshold > width)
    return fail(IR_ERR_INVALID_ARG, "threshold %d outside [1, min(H,W)/2]", threshold);
  if (x_plane_stride < (int64_t)height * width || out_plane_stride < (int64_t)height * width)
    return fail(IR_ERR_INVALID_ARG, "plane stride smaller than a plane");
  if ((planes + 3) / 4 > 0x7fffffffLL) return fail(IR_ERR_UNSUPPORTED, "grid too large");
  const hipError_t e = ir_launch_freeu_fourier(x, out, dtype, planes, height, width, x_plane_stride, out_plane_stride,
                                               threshold, scale, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "freeu launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_linear_fwd(int32_t dtype, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                  int64_t w_ld, const void* bias, void* y, int64_t y_ld, void* stream) {
  return ir_linear_fwd_scaled(dtype, 0, m, n, k, x, x_ld, w, w_ld, bias, y, y_ld, 0, 1.0f, stream);
}

int ir_linear_fwd_scaled(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                         int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                         void* stream) {
  return ir_linear_fwd_ex(dtype, x_is_f32, m, n, k, x, x_ld, w, w_ld, bias, y, y_ld, scale_cols, col_scale, IR_LIN_AUTO, stream);
}

static bool x_stationary_covers(int32_t n, int32_t k, const void* bias) {
  return ((k % 64 == 0 && k <= 320) || k == 640) && n % 32 == 0 && (bias == nullptr || n <= kLinearMaxBiasN);
}

int ir_linear_kernel_for(int64_t m, int32_t n, int32_t k, int32_t has_bias) {
  static const int dummy = 0;
  const bool xs = x_stationary_covers(n, k, has_bias ? (const void*)&dummy : nullptr);
  const bool tiled = (k % 64 == 0) && (n % 64 == 0);
  if (!xs && !tiled) return -1;
  if (!tiled) return IR_LIN_X_STATIONARY;
  // the X-stationary kernels read X once (and cast fp32 activations once) and win where that is the traffic that
  // matters: large M.  From profiles/r3_gemm_probe_final.txt: at 131072 rows they lead (K = 320: 122 vs 167 us), at 32768
  // rows the 256x256 tile leads or ties (32768 x 960 x 320: 33 vs 43 us; x 1920 x 640: 100 vs 100; x 640 x 640: 39-46 vs 44-54)
  if (xs && m >= 65536) return IR_LIN_X_STATIONARY;
  return IR_LIN_TILED_FIRST + ir_linear_tiled_pick(m, n, k);
}

static int linear_fwd_impl(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                           int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                           int32_t kernel, int32_t st_col0, int32_t st_cols, float* st_ws, size_t st_ws_bytes, void* stream);

int ir_linear_fwd_ex(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                     int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                     int32_t kernel, void* stream) {
  return linear_fwd_impl(dtype, x_is_f32, m, n, k, x, x_ld, w, w_ld, bias, y, y_ld, scale_cols, col_scale, kernel, 0, 0, nullptr, 0, stream);
}

static int stats_rows_of(int kernel) {   // every projection kernel gives a wave 64 rows of Y: the statistics block
  if (kernel == IR_LIN_X_STATIONARY) return 64;
  if (kernel >= IR_LIN_TILED_FIRST && kernel < IR_LIN_TILED_FIRST + IR_LIN_TILE_COUNT) return 64;
  return 0;
}

int ir_linear_stats_rows(int64_t m, int32_t n, int32_t k, int32_t has_bias) {
  if (m <= 0 || n <= 0 || k <= 0 || (n % 64) != 0) return 0;
  const int rows = stats_rows_of(ir_linear_kernel_for(m, n, k, has_bias));
  return (rows > 0 && (m % rows) == 0) ? rows : 0;
}

int ir_linear_fwd_stats(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                        int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                        int32_t stats_col0, int32_t stats_cols, float* stats_ws, size_t stats_ws_bytes, void* stream) {
  if (stats_ws == nullptr) return fail(IR_ERR_INVALID_ARG, "stats_ws is NULL (use ir_linear_fwd_scaled for a call without statistics)");
  return linear_fwd_impl(dtype, x_is_f32, m, n, k, x, x_ld, w, w_ld, bias, y, y_ld, scale_cols, col_scale, IR_LIN_AUTO, stats_col0, stats_cols,
                         stats_ws, stats_ws_bytes, stream);
}

int ir_adain_affine_from_partials(int32_t batch, int32_t heads, int32_t n_refs, int32_t len_self, int32_t len_ref,
                                  const float* style_ws, int32_t style_rows, const float* content_ws, int32_t content_rows,
                                  const float* content_mean, const float* content_std, const int32_t* valid, float eps,
                                  float* a, float* b, void* stream) {
  if (batch <= 0 || heads <= 0 || n_refs <= 0 || len_self <= 0 || len_ref <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!style_ws || !a || !b) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (style_rows <= 0 || (len_self % style_rows) != 0) return fail(IR_ERR_INVALID_ARG, "len_self %d is not a multiple of style_rows %d", len_self, style_rows);
  if (content_ws != nullptr) {
    if (content_rows <= 0 || (len_ref % content_rows) != 0) return fail(IR_ERR_INVALID_ARG, "len_ref %d is not a multiple of content_rows %d", len_ref, content_rows);
  } else if (!content_mean || !content_std) {
    return fail(IR_ERR_INVALID_ARG, "content statistics: either content_ws or content_mean + content_std");
  }
  if (len_self / style_rows > ir_adain_partials_max_chunks() || (content_ws != nullptr && len_ref / content_rows > ir_adain_partials_max_chunks()))
    return fail(IR_ERR_UNSUPPORTED, "more than %d partials per matrix (token axis too long for the merge kernel: use ir_adain_stats)", ir_adain_partials_max_chunks());
  AdainPartialsKParams p;
  memset(&p, 0, sizeof(p));
  p.style_ws = style_ws; p.content_ws = content_ws; p.cmean = content_mean; p.cstd = content_std; p.valid = valid;
  p.a = a; p.b = b; p.B = batch; p.H = heads; p.N = n_refs; p.Ls = len_self; p.Lr = len_ref;
  p.style_rows = style_rows; p.content_rows = content_ws != nullptr ? content_rows : 1; p.eps = eps;
  const hipError_t e = ir_launch_adain_affine_partials(p, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "adain_affine_partials launch: %s", hipGetErrorString(e));
  return IR_OK;
}

int ir_token_stats_from_partials(int32_t n_sets, int32_t heads, int32_t len, const float* ws, int32_t rows, float* mean, float* std,
                                 void* stream) {
  if (n_sets <= 0 || heads <= 0 || len <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (!ws || !mean || !std) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (rows <= 0 || (len % rows) != 0) return fail(IR_ERR_INVALID_ARG, "len %d is not a multiple of rows %d", len, rows);
  if (len / rows > ir_adain_partials_max_chunks()) return fail(IR_ERR_UNSUPPORTED, "more than %d partials per matrix", ir_adain_partials_max_chunks());
  const hipError_t e = ir_launch_token_stats_partials(ws, rows, n_sets, heads, len, mean, std, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "token_stats_partials launch: %s", hipGetErrorString(e));
  return IR_OK;
}

static int linear_fwd_impl(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                           int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                           int32_t kernel, int32_t st_col0, int32_t st_cols, float* st_ws, size_t st_ws_bytes, void* stream) {
  if (scale_cols < 0 || scale_cols > n || (scale_cols % 32) != 0) return fail(IR_ERR_INVALID_ARG, "scale_cols %d: a multiple of 32 in [0, N]", scale_cols);
  if (dtype != IR_DTYPE_F16 && dtype != IR_DTYPE_BF16) return fail(IR_ERR_UNSUPPORTED, "dtype %d: fp16 (0) / bf16 (1)", dtype);
  if (!x || !w || !y) return fail(IR_ERR_INVALID_ARG, "NULL pointer");
  if (m <= 0 || n <= 0 || k <= 0) return fail(IR_ERR_INVALID_ARG, "sizes must be > 0");
  if (kernel == IR_LIN_AUTO) kernel = ir_linear_kernel_for(m, n, k, bias != nullptr);
  if (kernel < 0) return fail(IR_ERR_UNSUPPORTED, "K = %d, N = %d: K must be a multiple of 64 and N of 64 (of 32 for K <= 320 or K = 640)", k, n);
  if (kernel == IR_LIN_X_STATIONARY) {
    if (!x_stationary_covers(n, k, bias))
      return fail(IR_ERR_UNSUPPORTED, "X-stationary kernel: K in {64,...,320} or 640, N %% 32 == 0, N <= %d with bias (K = %d, N = %d)", kLinearMaxBiasN, k, n);
  } else if (kernel < IR_LIN_TILED_FIRST || kernel >= IR_LIN_TILED_FIRST + IR_LIN_TILE_COUNT) {
    return fail(IR_ERR_INVALID_ARG, "kernel %d: 0 auto, 1 X-stationary, %d..%d tiled", kernel, IR_LIN_TILED_FIRST, IR_LIN_TILED_FIRST + IR_LIN_TILE_COUNT - 1);
  } else if (k % 64 != 0 || !ir_linear_tiled_cfg_ok(kernel - IR_LIN_TILED_FIRST, n)) {
    return fail(IR_ERR_UNSUPPORTED, "tiled kernel %d: K %% 64 == 0 and N a multiple of the tile width (K = %d, N = %d)", kernel, k, n);
  }
  if (m > 0x7fffffffLL - 256) return fail(IR_ERR_UNSUPPORTED, "M too large");
  if (x_ld < k || w_ld < k || y_ld < n || (x_ld % 8) || (w_ld % 8) || (y_ld % 8))
    return fail(IR_ERR_UNSUPPORTED, "leading dimensions must cover a row and be multiples of 8 elements");
  if (!aligned16(x) || !aligned16(w) || !aligned16(y) || (bias != nullptr && !aligned16(bias)))
    return fail(IR_ERR_UNSUPPORTED, "pointers must be 16-byte aligned");
  if ((int64_t)n * w_ld * 2 >= (1LL << 31)) return fail(IR_ERR_UNSUPPORTED, "weight larger than 2 GiB");
  if (256 * x_ld * 2 >= (1LL << 31)) return fail(IR_ERR_UNSUPPORTED, "x_ld too large");
  if (256 * y_ld * 2 >= (1LL << 31)) return fail(IR_ERR_UNSUPPORTED, "y_ld too large");   // a wave's 64 rows of Y sit behind one 32-bit buffer range
  LinearKParams p;
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.x_ld = x_ld; p.w_ld = w_ld; p.y_ld = y_ld;
  p.M = (int32_t)m; p.N = n; p.K = k; p.nsplit = 1;
  p.scale_cols = scale_cols; p.col_scale = col_scale; p.x_f32 = x_is_f32 ? 1 : 0;
  p.st_ws = nullptr; p.st_col0 = 0; p.st_cols = 0;
  if (st_ws != nullptr) {   // token statistics of columns [st_col0, st_col0 + st_cols) as the kernel's tail
    const int rows = stats_rows_of(kernel);
    if (st_col0 < 0 || st_cols <= 0 || (st_col0 % 64) != 0 || (st_cols % 64) != 0 || st_col0 + st_cols > n || (n % 64) != 0)
      return fail(IR_ERR_INVALID_ARG, "statistics columns [%d, %d + %d): whole heads (multiples of 64) inside N = %d, N %% 64 == 0", st_col0, st_col0, st_cols, n);
    if (rows <= 0 || (m % rows) != 0) return fail(IR_ERR_UNSUPPORTED, "statistics tail: M = %lld is not a multiple of the kernel's %d-row blocks (ir_linear_stats_rows)", (long long)m, rows);
    const size_t need = (size_t)(m / rows) * (size_t)(st_cols / 64) * 128 * sizeof(float);
    if (st_ws_bytes < need) return fail(IR_ERR_WORKSPACE, "stats_ws %zu < %zu bytes", st_ws_bytes, need);
    // the K = 640 X-stationary kernel re-reads the rows it has just stored (ir_wave_col_stats): every 128-byte line of the
    // statistics columns must belong to ONE workgroup's column range, i.e. rows of y start on a line and so does column st_col0
    if (kernel == IR_LIN_X_STATIONARY && k == 640 && (((y_ld * 2) % 128) != 0 || (((uintptr_t)y + (uintptr_t)st_col0 * 2) % 128) != 0))
      return fail(IR_ERR_UNSUPPORTED, "statistics tail at K = 640: y + stats_col0 must be 128-byte aligned and y_ld a multiple of 64 elements");
    p.st_ws = st_ws; p.st_col0 = st_col0; p.st_cols = st_cols;
  }
  const hipError_t e =
      kernel == IR_LIN_X_STATIONARY ? ir_launch_linear_skinny(p, dtype, (hipStream_t)stream)
                                    : ir_launch_linear_tiled(p, dtype, kernel - IR_LIN_TILED_FIRST, (hipStream_t)stream);
  if (e != hipSuccess) return fail(IR_ERR_LAUNCH, "linear launch: %s", hipGetErrorString(e));
  return IR_OK;
}

}  // extern "C"
