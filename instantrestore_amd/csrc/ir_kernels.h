// ir_kernels.h - kernel parameter blocks and host-side launchers (internal; the public
// interface is include/instantrestore_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define IR_KV_TILE 64          // keys per K/V tile
#define IR_ADAIN_ROWS 256      // token rows per AdaIN partial-statistics workgroup

// Parameter block of the fused attention kernels (passed by value as the kernel argument).
struct AttnKParams {
  const void* q;
  const void* k_self;
  const void* v_self;
  const void* k_ref;
  const void* v_ref;
  const float* aa;  // AdaIN scale  (B,N,H,64) or nullptr
  const float* ab;  // AdaIN shift
  void* out;
  float* lse;
  void* probs;      // attn_probs kernel only
  const int32_t* valid;   // ABI v8 (ir_shared_attn_args.valid_refs): references n >= valid[b] are all-zero in k_ref / v_ref: the default
                          // kernels (64-row, pipelined 32-row) close them analytically instead of walking their tiles; nullptr: walk all
  int64_t q_sb, q_sl, q_sh;
  int64_t ks_sb, ks_sl, ks_sh;
  int64_t vs_sb, vs_sl, vs_sh;
  int64_t kr_sb, kr_sn, kr_sl, kr_sh;
  int64_t vr_sb, vr_sn, vr_sl, vr_sh;
  int64_t o_sb, o_sl, o_sh;
  int B, H, Lq, Ls, N, Lr;
  int include_self;   // 0/1
  int q_prescaled;    // IR_FLAG_Q_PRESCALED: q holds Q * scale * log2(e)   (the 64-row kernel's QS instantiation, the 32-row kernel's PRESC forms)
  int out_f32;        // IR_FLAG_OUT_F32: fp32 output, o_s* in fp32 elements (every product kernel + combine)
  int tiles_self;     // ceil(Ls/64) if include_self else 0
  int tiles_ref;      // ceil(Lr/64)
  int ntiles;         // include_self*tiles_self + N*tiles_ref
  int nqb;            // query blocks per (b,h)
  int lkv;            // include_self*Ls + N*Lr
  float scale;
  float scale_log2;   // scale * log2(e)
  // remainder split (pipelined kernel): each XCD owns sk_ix consecutive work items; the first
  // sk_full of them (whole rounds of the XCD's workgroup slots) run their full K/V range, each
  // of the remaining sk_ix - sk_full is cut into sk_k pieces along the K/V tiles whose partial
  // (O, m, l) go to ws_o / ws_ml and are merged by the combine kernel.  sk_k <= 1: no split.
  float* ws;          // caller's workspace (or nullptr: never split)
  size_t ws_bytes;
  float* ws_o;        // [piece][QB rows][64] fp32, unnormalised O relative to m
  float* ws_ml;       // [piece][QB rows][2]  (raw running max, row sum)
  int sk_items;       // B*H*nqb
  int sk_ix;          // ceil(sk_items / 8)
  int sk_full;
  int sk_k;
  // ABI v9 (ir_shared_attn_args.seg_mass): attention mass per K/V segment as a BY-PRODUCT of the forward kernels.  At every segment
  // boundary of its K/V walk a kernel already holds the row sums (the AdaIN fold needs them); it stores the CUMULATIVE log-sum-exp
  // through segment s - a number that does not depend on the running reference - into seg_cum[b,h,row,s] (natural log; pieces of a
  // split item: log2 units into ws_cum, merged by the combine kernel), and seg_mass_finish_kernel turns the S cumulative values of
  // a row into S masses in place.  nullptr: off (nothing but one uniform test per SEGMENT, none per tile).
  float* seg_cum;     // (B, H, Lq, nseg_out) fp32 or nullptr
  float* ws_cum;      // [piece][QB rows][nseg_out]
  int nseg_out;       // include_self + N
};

struct AdainKParams {
  const void* v_self;
  const void* v_ref;
  int64_t vs_sb, vs_sl, vs_sh;
  int64_t vr_sb, vr_sn, vr_sl, vr_sh;
  float* ws;          // partial statistics: [B*(1+N)][H][nchunk][2][64] (mean, M2)
  float* a;
  float* b;
  int B, H, Ls, N, Lr;
  int nchunk;         // max(ceil(Ls/ROWS), ceil(Lr/ROWS))
  float eps;
  // ir_adain_stats_cached: content statistics (mean, unbiased std) of every reference V, (B, N, H, 64) fp32, computed once
  // per identity (ir_token_stats in the K/V-capture layer); only V_self is read here
  const float* cmean;
  const float* cstd;
};

struct AdainApplyKParams {
  const void* x;
  void* y;
  const float* a;
  const float* b;
  int64_t x_sb, x_sn, x_sl, x_sh;
  int64_t y_sb, y_sn, y_sl, y_sh;
  int B, H, N, L;
};

struct ZeroRefsKParams {
  void* k;
  void* v;
  const int32_t* valid;
  int64_t k_sb, k_sn, k_sl, k_sh;
  int64_t v_sb, v_sn, v_sl, v_sh;
  int B, H, N, L;
};

// launchers (defined next to their kernels); dtype: 0 = f16, 1 = bf16. Return hipError_t.
hipError_t ir_launch_shared_attn_fwd(const AttnKParams& p, int dtype, int variant, hipStream_t s);
hipError_t ir_launch_shared_attn_fwd_pipe(const AttnKParams& p, int dtype, int nw, hipStream_t s);
hipError_t ir_launch_shared_attn_fwd_pipe_abl(const AttnKParams& p, int abl, hipStream_t s);
hipError_t ir_launch_shared_attn_combine(const AttnKParams& p, int dtype, int qb, int rem, hipStream_t s);
hipError_t ir_launch_shared_attn_fwd_w64(const AttnKParams& p, int dtype, hipStream_t s);
hipError_t ir_launch_shared_attn_fwd_w64x8(const AttnKParams& p, int dtype, hipStream_t s);
// round 6: one wave per SIMD, 128 query rows per wave, hand-placed instruction stream (shared_attn_fwd_w128.hip)
hipError_t ir_launch_shared_attn_fwd_w128(const AttnKParams& p, int dtype, hipStream_t s);
bool ir_attn_w128_supports(const AttnKParams& p);   // pre-scaled Q, whole 64-key tiles, no valid_refs / seg_mass
bool ir_attn_default_is_w128(const AttnKParams& p);
#ifdef IR_ABLATIONS   // energy / timing ablations of the 64-row QS kernel (tuning values 20 + index; shared_attn_fwd_w64.hip)
hipError_t ir_launch_shared_attn_fwd_w64_abl(const AttnKParams& p, int dtype, int index, hipStream_t s);
int ir_w64_abl_count(void);
int ir_w64_abl_mask(int index);
#endif
hipError_t ir_launch_seg_mass_finish(const AttnKParams& p, hipStream_t s);   // cumulative log-sum-exp -> masses, in place
bool ir_attn_default_is_w64(const AttnKParams& p);
bool ir_attn_variant_available(int variant);

// Remainder split: `rem` items of the last, partially filled round (per XCD) on `slots` concurrently
// resident workgroups.  Cutting each into k K/V-range pieces makes the round last ceil(rem*k/slots)/k of an
// item; pick the k that minimises it (plus a small per-piece charge for the fp32 partials and the combine),
// within the piece-length floor `kmax` and the workspace capacity `cap_pieces` (pieces per XCD).
static inline int ir_pick_split(int rem, int slots, int kmax, long cap_pieces) {
  int best_k = 1;
  double best = 1.0;   // k = 1: one round of whole items
  for (int k = 2; k <= kmax && (long)rem * k <= cap_pieces; ++k) {
    const int rounds = (rem * k + slots - 1) / slots;
    const double t = (double)rounds / k + 0.012 * k;
    if (t < best - 1e-9) { best = t; best_k = k; }
  }
  return best_k;
}   // the default dispatch rule (variant 0)
// variant: 0 = automatic (line kernel when every segment length is a multiple of 8), 1 = round 1's 2-byte-store kernel,
// 2 / 3 = the line kernel with 64 / 32 query rows per wave
hipError_t ir_launch_attn_probs(const AttnKParams& p, int dtype, int variant, hipStream_t s);
bool ir_attn_probs_uses_lines(const AttnKParams& p);
hipError_t ir_launch_attn_segment_mass(const AttnKParams& p, int dtype, float* mass, hipStream_t s);
hipError_t ir_launch_adain_stats(const AdainKParams& p, int dtype, hipStream_t s);
hipError_t ir_launch_token_stats(const AdainKParams& p, int dtype, hipStream_t s);
hipError_t ir_launch_adain_stats_cached(const AdainKParams& p, int dtype, hipStream_t s);
// round 4: the affine / the plain token statistics from the partials the projection GEMMs leave behind (ir_colstats.h)
struct AdainPartialsKParams {
  const float* style_ws;     // partials of V_self: [(b * Ls / style_rows + c)][H][128]
  const float* content_ws;   // partials of the reference V's: [((b * N + n) * Lr / content_rows + c)][H][128], or nullptr
  const float* cmean;        // ... then the finished content statistics (B, N, H, 64): mean, unbiased std (no eps)
  const float* cstd;
  const int32_t* valid;      // optional (B): references n >= valid[b] were zero-filled: statistics (0, 0)
  float* a;
  float* b;
  int B, H, N, Ls, Lr, style_rows, content_rows;
  float eps;
};
hipError_t ir_launch_adain_affine_partials(const AdainPartialsKParams& p, hipStream_t s);
hipError_t ir_launch_token_stats_partials(const float* ws, int rows, int nsets, int H, int len, float* mean, float* std, hipStream_t s);
// benchmark hook (bench_hooks.hip): one launch of an MFMA-only stream, `blocks` workgroups x 8 waves x iters x 16 MFMAs
hipError_t ir_launch_bench_mfma_stream(int dtype, int zero, int iters, int blocks, float* out, hipStream_t s);
int ir_adain_partials_max_chunks(void);   // partials per matrix the merge kernels hold in registers (len / rows <= this)
hipError_t ir_launch_adain_apply(const AdainApplyKParams& p, int dtype, hipStream_t s);
hipError_t ir_launch_zero_refs(const ZeroRefsKParams& p, hipStream_t s);
hipError_t ir_launch_tensor2im(const void* x, void* out, int dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                               int B, int C, int H, int W, hipStream_t s);

// ---- image_io.hip: the caller's input transform and FreeU's skip-feature filter ----------------
struct ResampleImageK {
  const unsigned char* src;   // (in_h, in_w, 3) uint8, pixel stride 3 bytes
  int64_t src_row_bytes;
  const int32_t* bounds_h;    // (out_w, 2): first source column, tap count
  const int32_t* kk_h;        // (ksize_h, out_w) 22-bit fixed-point taps, TRANSPOSED (tap-major)
  const int32_t* bounds_v;    // (out_h, 2)
  const int32_t* kk_v;        // (out_h, ksize_v)
  unsigned char* tmp;         // (row_count, size, 3): horizontal pass of the crop's columns
  int32_t ksize_h, ksize_v;
  int32_t crop_top, crop_left;
  int32_t row_first, row_count;  // source rows the crop's vertical taps touch
  int32_t col_first, col_count;  // source columns the crop's horizontal taps touch
  int32_t in_h, in_w, out_w;
};
constexpr int kPreprocessImagesPerLaunch = 16;
struct PreprocessKParams {
  ResampleImageK img[kPreprocessImagesPerLaunch];
  int32_t n, size, first_image, tmp_pitch;  // tmp rows are tmp_pitch bytes apart (multiple of 4)
};
hipError_t ir_launch_preprocess(const PreprocessKParams& p, int max_rows, int max_span_bytes, int max_ksize_v, int dtype,
                                void* out, hipStream_t s);
hipError_t ir_launch_freeu_fourier(const void* x, void* out, int dtype, int64_t planes, int H, int W, int64_t sp_in,
                                   int64_t sp_out, int thr, float scale, hipStream_t s);
int ir_host_lanczos_ksize(int in_size, int out_size);

// ---- linear_skinny.hip: Y = X W^T (+ bias) for K <= 320 (X-stationary, W streamed) ---------------
constexpr int kLinearMaxBiasN = 4096;
struct LinearKParams {
  const void* x;     // (M, K) rows x_ld elements apart
  const void* w;     // (N, K) rows w_ld elements apart (torch Linear weight)
  const void* bias;  // (N) or nullptr
  void* y;           // (M, N) rows y_ld elements apart
  int64_t x_ld, w_ld, y_ld;
  int32_t M, N, K, nsplit;
  int32_t x_f32;        // x is fp32 (x_ld in fp32 elements): cast to the 16-bit type while loading the resident fragments
  int32_t scale_cols;   // output columns [0, scale_cols) are multiplied by col_scale in fp32 before the rounding
  float col_scale;      // (scale_cols % 32 == 0; 0 = none): the softmax scale * log2(e) on the q third of a fused q/k/v
  // round 4: token statistics of a column range (the V third of a fused q/k/v output) as the kernel's tail (ir_colstats.h):
  // one (mean[64], M2[64]) partial per (64-row block, head); ws == nullptr: off
  float* st_ws;
  int32_t st_col0, st_cols;
};
hipError_t ir_launch_linear_skinny(const LinearKParams& p, int dtype, hipStream_t s);

// ---- linear_tiled.hip: LDS-tiled Y = X W^T (+ bias) for any K % 64 == 0, N % 64 == 0 (K = 1280, small-M shapes) ----
enum {   // tile shapes (rows x columns of Y per workgroup); values are the `kernel` argument of ir_linear_fwd_ex minus 2
  IR_LIN_TILE_256x128 = 0,   // 8 waves
  IR_LIN_TILE_128x128 = 1,   // 4 waves
  IR_LIN_TILE_128x64 = 2,    // 2 waves
  IR_LIN_TILE_256x64 = 3,    // 4 waves
  IR_LIN_TILE_64x128 = 4,    // 2 waves
  IR_LIN_TILE_128x256 = 5,   // 4 waves, 64 x 128 per wave
  IR_LIN_TILE_256x256 = 6,   // 8 waves, 64 x 128 per wave, wave groups one phase apart (matrix phase beside load phase on every SIMD)
  IR_LIN_TILE_128x128_K2 = 7,   // round 5: 128 x 128 with the contraction split over two 4-wave groups (K / 64 even), fixed-order sum through LDS
  IR_LIN_TILE_COUNT = 8
};
hipError_t ir_launch_linear_tiled(const LinearKParams& p, int dtype, int cfg, hipStream_t s);
int ir_linear_tiled_pick(int64_t M, int N, int K);
bool ir_linear_tiled_cfg_ok(int cfg, int N);
void ir_host_lanczos_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk);
