/*
 * instantrestore_hip.h - C ABI of the MI355X (gfx950) hot path of InstantRestore.
 *
 * This is the drop-in boundary of the build: a plain-C interface (raw device pointers,
 * explicit strides, a hipStream_t passed as void*) that the Python host side binds with
 * ctypes (instantrestore_amd/_lib.py) and that a maintainer of the reference would bind from
 * face_replace/models/attn_processors.py.  The reference itself is pure
 * PyTorch and has no FFI; each entry point below names the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success and a negative ir_status on failure; nothing is
 *     thrown across the ABI; ir_last_error_string() describes the last failure of the
 *     calling thread;
 *   - all device work is enqueued asynchronously on `stream` (a hipStream_t; NULL = the
 *     legacy default stream); nothing synchronises, nothing allocates;
 *   - the caller owns every buffer; the library keeps no state between calls (no process-wide
 *     switches: kernel selection for benchmarks is the per-call `tuning` field of the args);
 *   - tensors are described by a base pointer plus strides IN ELEMENTS; the innermost
 *     (head_dim) axis is always contiguous and head_dim is always IR_HEAD_DIM = 64
 *     (SD-Turbo: attention_head_dim [5,10,20,20] x 64 channels);
 *   - element type of q/k/v/out is selected by ir_dtype (fp16 or bf16, 2 bytes); softmax
 *     statistics, AdaIN statistics and all accumulation are fp32.
 */
#ifndef INSTANTRESTORE_HIP_H
#define INSTANTRESTORE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IR_ABI_VERSION 9
#define IR_HEAD_DIM 64

typedef enum ir_status {
  IR_OK = 0,
  IR_ERR_INVALID_ARG = -1, /* NULL pointer, negative size, bad stride, bad struct_size */
  IR_ERR_UNSUPPORTED = -2, /* dtype / head_dim / alignment the kernels do not implement */
  IR_ERR_LAUNCH = -3,      /* hipLaunchKernel reported an error (see ir_last_error_string) */
  IR_ERR_WORKSPACE = -4    /* workspace too small */
} ir_status;

typedef enum ir_dtype { IR_DTYPE_F16 = 0, IR_DTYPE_BF16 = 1 } ir_dtype;

/* flags of ir_shared_attn_args.flags */
#define IR_FLAG_INCLUDE_SELF 1u /* train_input: self K/V block precedes the reference blocks */
#define IR_FLAG_Q_PRESCALED 2u  /* q already holds Q * scale * log2(e) (the fused q/k/v projection applies the factor to its
                                   fp32 accumulator before the one rounding to 16 bit, ir_linear_fwd col_scale): the
                                   scores leave the matrix pipe in the exp2 domain and the kernel saves one multiply-add
                                   per score.  `scale` is still the reference's attn.scale (used for the LSE). */
#define IR_FLAG_OUT_F32 4u      /* out is fp32 (strides in fp32 elements): the result of the SAME kernel the call would launch
                                   otherwise, stored BEFORE its rounding to the 16-bit type - parity instrumentation
                                   (tests show the pre-rounding error of the kernel that ships) */

/*
 * ir_shared_attn_fwd - fused extended self-attention (flash-style, no probability matrix).
 *
 * Replaces the body of SharedAttnProcessor.forward between the q/k/v projections and to_out
 * (face_replace/models/attn_processors.py:232-264): the 3+2N head_to_batch_dim copies
 * (:232-241), adain() of every reference V (:242-246, when adain_a/adain_b are given), the
 * torch.cat of the extended K/V (:247-252), get_attention_scores = baddbmm+softmax (:257,
 * diffusers 0.24 Attention) and the bmm + batch_to_head_dim (:263-264).  With n_refs == 0 it is
 * the plain attention of AttnProcessor.forward (:76-82) and of the self_attn_idx=None branch
 * (:253-255), cross attention included (kv length = len_self).
 *
 *   O[b,i,h,:] = sum_j softmax_j( scale * <Q[b,i,h,:], K_ext[b,j,h,:]> ) * V_ext[b,j,h,:]
 *   K_ext / V_ext column order: [self tokens (iff IR_FLAG_INCLUDE_SELF)] ++ ref 0 ++ ... ++ ref N-1
 *   V_ext of reference n is  v_ref[b,n,j,h,d] * adain_a[b,n,h,d] + adain_b[b,n,h,d]  (if given).
 *
 * Layouts (strides in elements, d contiguous):
 *   q      (B, len_q,   H, 64)  strides q_sb, q_sl, q_sh
 *   k_self (B, len_self,H, 64)  strides ks_sb, ks_sl, ks_sh      v_self: vs_*
 *   k_ref  (B, N, len_ref, H, 64) strides kr_sb, kr_sn, kr_sl, kr_sh   v_ref: vr_*
 *   out    (B, len_q,   H, 64)  strides o_sb, o_sl, o_sh
 *   adain_a, adain_b : fp32 (B, N, H, 64) contiguous, or both NULL
 *   lse    : fp32 (B, H, len_q) contiguous or NULL; natural-log sum-exp of the scaled scores
 * i.e. a (B, L, C=H*64) activation is passed as is with sl = C, sh = 64, and the reference's
 * ref_keys[idx] (B, N, L, C) with sn = L*C (pix2pix_turbo.py:265-266): nothing is copied.
 * All base pointers and strides must keep every (row, head) vector 16-byte aligned.
 */
typedef struct ir_shared_attn_args {
  uint32_t struct_size; /* = sizeof(ir_shared_attn_args) */
  int32_t dtype;        /* ir_dtype */
  uint32_t flags;       /* IR_FLAG_* */
  int32_t batch;        /* B */
  int32_t heads;        /* H */
  int32_t len_q;        /* query tokens */
  int32_t len_self;     /* tokens of k_self / v_self (0 allowed iff !INCLUDE_SELF) */
  int32_t n_refs;       /* N (0 = plain attention) */
  int32_t len_ref;      /* tokens per reference */
  float scale;          /* attn.scale = head_dim ** -0.5 */
  const void* q;
  const void* k_self;
  const void* v_self;
  const void* k_ref;
  const void* v_ref;
  const float* adain_a;
  const float* adain_b;
  void* out;
  float* lse;
  int64_t q_sb, q_sl, q_sh;
  int64_t ks_sb, ks_sl, ks_sh;
  int64_t vs_sb, vs_sl, vs_sh;
  int64_t kr_sb, kr_sn, kr_sl, kr_sh;
  int64_t vr_sb, vr_sn, vr_sl, vr_sh;
  int64_t o_sb, o_sl, o_sh;
  void* workspace;          /* optional device scratch (see ir_shared_attn_workspace_bytes), or NULL */
  uint64_t workspace_bytes;
  int32_t tuning;           /* 0 = default kernel dispatch.  Benchmarks / A-B tests only: IR_TUNE_* selects one kernel
                               for this call (an unknown or unavailable value is IR_ERR_UNSUPPORTED). */
  int32_t reserved;         /* must be 0 */
  const int32_t* valid_refs; /* ABI v8, optional: int32 (B) on the device.  References n >= valid_refs[b] of batch entry b are
                               ALL-ZERO in k_ref and v_ref (ir_zero_invalid_refs; pix2pix_turbo.py:269-273) - a promise by the
                               caller, which lets the kernel account for those segments in closed form (every score of a zero
                               key is exactly 0, every value row is 0, or the AdaIN shift b when an affine is given) instead of
                               walking their tiles.  Same result as with NULL: zeroed, not masked - the tokens keep their exp(0)
                               weight.  NULL = every reference is walked. */
  float* seg_mass;          /* ABI v9, optional output: fp32 (B, H, len_q, S) contiguous, S = (INCLUDE_SELF ? 1 : 0) + n_refs - the
                               attention mass of every K/V segment, mass[b,h,i,s] = sum over the keys j of segment s of
                               softmax_j(scale * <q_i, k_j>), as a BY-PRODUCT of this launch: the kernels hold the row sums at
                               every segment boundary anyway and store the cumulative log-sum-exp there; a row-sized finishing
                               kernel on the same stream turns them into masses (the S masses of a row sum to exactly 1).  What
                               gradio_demo.py:119-127 reduces attention_probs to, with no second pass over Q and K
                               (ir_attn_segment_mass is that second pass, for callers that only kept the LSE).  Taken as
                               differences of cumulative sums: absolute accuracy ~1e-6, not relative.  Column order: segment 0
                               is the SELF segment when INCLUDE_SELF is set, so the reference demo's blocks idx 0..3 (which start
                               at column 0) are mass[..., 0:4] - valid only when len_self == len_ref - and the per-REFERENCE
                               masses are mass[..., INCLUDE_SELF:].  NULL = off. */
} ir_shared_attn_args;

/* values of ir_shared_attn_args.tuning (csrc/shared_attn_fwd.hip lists what each one is) */
#define IR_TUNE_DEFAULT 0
#define IR_TUNE_PIPE32_EXACTMAX 7
#define IR_TUNE_PIPE32 10
#define IR_TUNE_PIPE32_PRESCALE_Q 11
#define IR_TUNE_W64X4 12
#define IR_TUNE_W64X8 13
#define IR_TUNE_PIPE32_EARLYQK 14
#define IR_TUNE_W128 16            /* round 6: one wave per SIMD, 128 query rows per wave, hand-placed instruction stream; needs
                                      IR_FLAG_Q_PRESCALED, segment lengths that are multiples of 64, no valid_refs / seg_mass */
#define IR_TUNE_W64_ABL_FIRST 20 /* 20 ... : development builds (-DIR_ABLATIONS) only - energy / timing ablations of the 64-row kernel
                                    (WRONG results: one class of work removed per bit; tools/gpu_energy_probe.py).  Values 16 / 17
                                    (rounds 2-3: one-wave-per-SIMD and three-stage experiments) are retired. */
#define IR_TUNE_PIPE32_POSTCHECK 18   /* 32-row kernel, pre-scaled Q, reference checked after the exponentials (needs
                                         IR_FLAG_Q_PRESCALED; parity-green, same speed as PIPE32_PRESCALE_Q: opt-in) */

/*
 * Scratch for the remainder split: when the number of (batch, head, query-block) work items is not
 * a multiple of the resident workgroup slots, the items of the last, partially filled round are
 * cut into K/V-range pieces whose partial (O, max, sum) results are merged by a second small
 * kernel, so that round ends early instead of running at full length.  Passing NULL (or a smaller
 * buffer) only disables (or limits) the split; results are identical up to fp32 rounding.
 */
size_t ir_shared_attn_workspace_bytes(void);

/* Name of the kernel ir_shared_attn_fwd would launch for these arguments (incl. their `tuning` field)
 * (reporting only: bench.py's roofline block); "" if the arguments are invalid. Static storage. */
const char* ir_shared_attn_kernel_name(const ir_shared_attn_args* args);

int ir_shared_attn_fwd(const ir_shared_attn_args* args, void* stream);

/*
 * ir_attn_probs - materialise attention_probs (B, H, len_q, Lkv) for the dump path.
 *
 * Replaces `self.attention_probs = attention_probs.reshape(B, heads, L, Lkv)`
 * (attn_processors.py:258-261), read by test.py:108, gradio_demo.py:118, coach.py:312.
 * Uses the same args as ir_shared_attn_fwd (out / adain_* are ignored) plus the `lse` that
 * call produced:  probs[b,h,i,j] = exp(scale*<q_i,k_j> - lse[b,h,i]), stored in `dtype`,
 * contiguous, column order as above.
 *
 * HBM-write bound (H*L*Lkv*2 bytes per identity).  When len_self and len_ref are multiples of 8 keys (every row of
 * probs then starts 16-byte aligned) the line kernel runs: whole 128-byte lines per store instruction (round 5); other
 * lengths take the 2-byte-store kernel.  ir_attn_probs_ex names the kernel (benchmarks / A-B tests).
 */
int ir_attn_probs(const ir_shared_attn_args* args, void* probs, void* stream);
#define IR_PROBS_AUTO 0
#define IR_PROBS_GENERIC 1   /* 2-byte stores, any length */
#define IR_PROBS_LINES64 2       /* line kernel, 64 query rows x 64 keys (128 B per row) per wave and step */
#define IR_PROBS_LINES32 3       /* 32 rows x 64 keys */
#define IR_PROBS_LINES32_K128 4  /* 32 rows x 128 keys (256 B per row and store batch) */
#define IR_PROBS_LINES64_K128 5  /* 64 rows x 128 keys */
#define IR_PROBS_LINES32_K256 6  /* 32 rows x 256 keys (512 B per row and store batch) */
int ir_attn_probs_ex(const ir_shared_attn_args* args, void* probs, int32_t kernel, void* stream);

/*
 * ir_attn_segment_mass (ABI v8, opt-in) - attention mass per K/V segment without the probability matrix.
 *
 *   mass[b,h,i,s] = sum over the keys j of segment s of exp(scale*<q_i,k_j> - lse[b,h,i]),  fp32 (B, H, len_q, S) contiguous,
 *   segments s in the column order above: [self (iff IR_FLAG_INCLUDE_SELF)] ++ ref 0 ++ ... ++ ref N-1, S = include_self + N.
 * What gradio_demo.py:119-127 reduces attention_probs to (`probs[..., attn_size*idx : attn_size*(idx+1)].sum(-1)` per
 * reference): the same numbers (summed in fp32 before any 16-bit rounding) without writing H*L*Lkv*2 bytes.  Same args as
 * ir_attn_probs; any segment length.
 */
int ir_attn_segment_mass(const ir_shared_attn_args* args, float* mass, void* stream);

/*
 * ir_adain_stats - per-(b, n, channel) AdaIN affine from token statistics.
 *
 * Replaces the statistics half of adain() (attn_processors.py:9-10) and of its call site
 * (:244-245): mean and UNBIASED standard deviation over the token axis of the degraded
 * image's V (style) and of every reference V (content), eps = 1e-5 added to both deviations:
 *      a = (std(v_self)+eps) / (std(v_ref_n)+eps),   b = mean(v_self) - mean(v_ref_n) * a
 * so that adain(v_ref_n) == v_ref_n * a + b.  One pass over the data, fp32 Chan/Welford merge.
 *   v_self (B, len_self, H, 64) strides vs_*;  v_ref (B, N, len_ref, H, 64) strides vr_*
 *   a, b : fp32 (B, N, H, 64) contiguous
 *   workspace: >= ir_adain_stats_workspace_bytes(...) bytes of device memory
 */
size_t ir_adain_stats_workspace_bytes(int32_t batch, int32_t heads, int32_t len_self, int32_t n_refs,
                                      int32_t len_ref);
int ir_adain_stats(int32_t dtype, int32_t batch, int32_t heads, int32_t len_self, int32_t n_refs,
                   int32_t len_ref, const void* v_self, int64_t vs_sb, int64_t vs_sl, int64_t vs_sh,
                   const void* v_ref, int64_t vr_sb, int64_t vr_sn, int64_t vr_sl, int64_t vr_sh,
                   float eps, float* a, float* b, void* workspace, size_t workspace_bytes, void* stream);
/*
 * ir_adain_stats_cached - the same affine when the CONTENT statistics are already known.
 *
 * mean(v_ref_n) and std(v_ref_n) of a reference V do not change while the identity's references do not
 * (pix2pix_turbo.py:255-266 recomputes them every frame): the K/V-capture layer emits them once with ir_token_stats
 * (content_mean, content_std: fp32 (B, N, H, 64) contiguous, std WITHOUT the eps) and every frame reads only v_self,
 * 1/(N+1) of the bytes.  Bit-identical (a, b) to ir_adain_stats on the same tensors: both run the same partial kernel
 * and the same merge order.  A reference zero-filled by ir_zero_invalid_refs has statistics (0, 0): the caller zeroes
 * its cached entries (instantrestore_amd/kv_harvest.py does).
 *   workspace: >= ir_adain_stats_workspace_bytes(batch, heads, len_self, 0, len_self)
 */
int ir_adain_stats_cached(int32_t dtype, int32_t batch, int32_t heads, int32_t len_self, int32_t n_refs,
                          const void* v_self, int64_t vs_sb, int64_t vs_sl, int64_t vs_sh,
                          const float* content_mean, const float* content_std,
                          float eps, float* a, float* b, void* workspace, size_t workspace_bytes, void* stream);

/*
 * ir_token_stats - mean and UNBIASED standard deviation over the token axis.
 *
 * The content-statistics lines of adain() on their own (attn_processors.py:9-10, without the
 * +1e-5): used by the exported adain(content, style_mean, style_std), whose style statistics
 * arrive precomputed.   x (B, M, len, H, 64) strides x_*;  mean, std: fp32 (B, M, H, 64).
 * workspace >= ir_adain_stats_workspace_bytes(batch, heads, len, n_mats - 1, len).
 */
int ir_token_stats(int32_t dtype, int32_t batch, int32_t heads, int32_t n_mats, int32_t len,
                   const void* x, int64_t x_sb, int64_t x_sn, int64_t x_sl, int64_t x_sh,
                   float* mean, float* std, void* workspace, size_t workspace_bytes, void* stream);

/*
 * ir_adain_apply - standalone AdaIN application  y = x * a + b  (fp32 math, stored in dtype).
 *
 * Replaces the normalise/scale/shift half of adain() (attn_processors.py:12-16) for callers
 * that need the renormalised V materialised (op-level parity, the exported adain()).  The fused
 * attention folds the same affine into its V staging and never writes this tensor.
 *   x, y (B, N, len, H, 64) with strides x_* / y_*;  a, b fp32 (B, N, H, 64) contiguous
 */
int ir_adain_apply(int32_t dtype, int32_t batch, int32_t heads, int32_t n_refs, int32_t len,
                   const void* x, int64_t x_sb, int64_t x_sn, int64_t x_sl, int64_t x_sh,
                   const float* a, const float* b,
                   void* y, int64_t y_sb, int64_t y_sn, int64_t y_sl, int64_t y_sh, void* stream);

/*
 * ir_zero_invalid_refs - zero the K and V of references n >= valid[b] in place.
 *
 * Replaces the Python double loop of Pix2Pix_Turbo.get_conditioning_keys_values
 * (face_replace/models/pix2pix_turbo.py:269-273).  Zeroed, not masked: the tokens keep their
 * exp(0) softmax weight, exactly like the reference.  valid: int32 (B) on the device.
 *   k, v (B, N, len, H*64) described as (B, N, len, H, 64) with strides *_sb,_sn,_sl,_sh
 */
int ir_zero_invalid_refs(int32_t batch, int32_t heads, int32_t n_refs, int32_t len, const int32_t* valid,
                         void* k, int64_t k_sb, int64_t k_sn, int64_t k_sl, int64_t k_sh,
                         void* v, int64_t v_sb, int64_t v_sn, int64_t v_sl, int64_t v_sh, void* stream);

/*
 * ir_tensor2im_u8 - the caller's output path on the device.
 *
 * Replaces tensor2im(var, unnorm=True) (face_replace/training/utils/vis_utils.py:14-23, called at
 * face_replace/inference/test.py:139): x*0.5+0.5 in the tensor's dtype, clamp [0,1], *255 in that
 * dtype, truncation to uint8, CHW -> HWC.  Bit-identical bytes; only H*W*C bytes cross PCIe.
 *   x (B, C, H, W) strides x_sb, x_sc, x_sh, x_sw (elements); dtype 0 = fp16, 1 = bf16, 2 = fp32
 *   out_u8 (B, H, W, C) contiguous uint8
 */
int ir_tensor2im_u8(int32_t dtype, int32_t batch, int32_t channels, int32_t height, int32_t width, const void* x,
                    int64_t x_sb, int64_t x_sc, int64_t x_sh, int64_t x_sw, void* out_u8, void* stream);

/*
 * ir_preprocess_lanczos_u8 - the caller's input transform on the device.
 *
 * Replaces, for a batch of differently sized RGB images, the per-image CPU pipeline of
 * face_replace/inference/test.py:54-59 (applied at :76, :127, :150):
 *     Resize(size, LANCZOS) -> CenterCrop(size) -> ToTensor() -> Normalize(0.5, 0.5)
 * i.e. Pillow's 8-bit two-pass resampler (pillow==10.4.0 Resample.c; horizontal pass rounded to
 * uint8, then vertical, 22-bit fixed-point taps) and (v/255 - 0.5)/0.5 in float32, cast to
 * out_dtype.  The resampled bytes are bit-identical to Pillow's; only the crop is computed.
 *
 * The tap tables are Pillow's precompute_coeffs + normalize_coeffs_8bpc, produced on the HOST by
 * ir_lanczos_ksize / ir_lanczos_coeffs (no GPU involved; same libm and operation order) and uploaded
 * by the caller once per (in_size, out_size); out sizes and crop offsets follow torchvision 0.15.2
 * (instantrestore_amd/preprocess.py).
 *   images: HOST array of n descriptors; every pointer inside is a DEVICE pointer
 *   out   : (n, 3, size, size) contiguous, dtype 0 = fp16, 1 = bf16, 2 = fp32
 */
typedef struct ir_image_desc {
  const void* src;          /* (in_h, in_w, 3) uint8, rows src_row_bytes apart */
  int64_t src_row_bytes;
  int32_t in_h, in_w;
  int32_t out_h, out_w;     /* resized size (before the crop) */
  int32_t crop_top, crop_left;
  const int32_t* bounds_h;  /* (out_w, 2) */
  const int32_t* kk_h;      /* (ksize_h, out_w): TAP-MAJOR, the transpose of ir_lanczos_coeffs' table */
  const int32_t* bounds_v;  /* (out_h, 2) */
  const int32_t* kk_v;      /* (out_h, ksize_v) */
  int32_t ksize_h, ksize_v;
  int32_t row_first, row_count; /* source rows touched by the vertical taps of the crop's rows */
  int32_t col_first, col_count; /* source columns touched by the horizontal taps of the crop's columns */
  void* tmp;                /* 4-byte aligned scratch, >= row_count * ((size*3 + 3) & ~3) bytes */
} ir_image_desc;

int ir_lanczos_ksize(int32_t in_size, int32_t out_size);            /* taps per output sample; < 0 on error */
int ir_lanczos_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds /* host (out,2) */,
                      int32_t* kk /* host (out, ksize) */);
int ir_preprocess_lanczos_u8(const ir_image_desc* images, int32_t n_images, int32_t size, int32_t out_dtype,
                             void* out, void* stream);

/*
 * ir_freeu_fourier_filter - FreeU's skip-feature filter.
 *
 * Replaces fourier_filter(res_hidden_states.float(), threshold, scale).to(dtype)
 * (face_replace/models/unet_2d_condition/block.py:3514,3518 -> diffusers==0.24.0
 * utils/torch_utils.py): fftn -> fftshift -> multiply the (2*threshold)^2 centre bins by `scale`
 * -> ifftshift -> ifftn -> real.  Computed in closed form (only those bins change): one read of x,
 * one write of out, fp32 arithmetic, no intermediate tensors.  In place (out == x) is allowed.
 *   x, out: `planes` = B*C planes of height*width contiguous elements, plane strides in elements
 *   dtype 0 = fp16, 1 = bf16, 2 = fp32; height*width <= 4096; 1 <= threshold <= min(H,W)/2
 */
int ir_freeu_fourier_filter(int32_t dtype, int64_t planes, int32_t height, int32_t width, const void* x,
                            int64_t x_plane_stride, void* out, int64_t out_plane_stride, int32_t threshold,
                            float scale, void* stream);

/*
 * ir_linear_fwd - y = x W^T (+ bias): the q/k/v (fused, N = 3C) and out projections of every layer class
 * (SURVEY.md section 8f rank 4).
 *
 * Replaces attn.to_q/to_k/to_v and attn.to_out[0] (nn.Linear; attn_processors.py:222-230,267).  Two kernel families,
 * both fp32 accumulation with ONE rounding to the 16-bit dtype (bias added in fp32 before it), both deterministic
 * (no atomics, no split-K reduction through memory):
 *   - X-stationary (K in {64, 128, ..., 320}, and K = 640 with the contraction split over two waves whose fp32
 *     partial tiles meet in LDS): X is read once and kept in registers, W streams through LDS; N % 32 == 0,
 *     N <= 4096 with bias.  Chosen for large M (M >= 65536 rows).
 *   - LDS-tiled (any K % 64 == 0, N % 64 == 0): 256x256 ... 64x128 tiles of Y, both operands through swizzled LDS
 *     stages; K = 1280 and the small-M shapes of every class.
 * Other shapes return IR_ERR_UNSUPPORTED.
 *   x (M, K) rows x_ld elements apart; w (N, K) rows w_ld apart (torch Linear weight layout);
 *   bias (N) or NULL; y (M, N) rows y_ld apart; all ld % 8 == 0, pointers 16-B aligned
 */
int ir_linear_fwd(int32_t dtype, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                  int64_t w_ld, const void* bias, void* y, int64_t y_ld, void* stream);
/* The same with (a) output columns [0, scale_cols) multiplied by col_scale in the fp32 accumulator, before the one rounding
 * (y = acc * col_scale + bias there): the fused q/k/v projection hands the attention kernel Q * scale * log2(e)
 * (IR_FLAG_Q_PRESCALED) at no extra rounding; scale_cols % 32 == 0; (b) x_is_f32 != 0: x is fp32 (x_ld in fp32
 * elements) and is rounded to `dtype` while it is loaded - the `.to(fp16)` the reference's autocast performs on the
 * LayerNorm output ahead of every projection (inference/test.py:83), without its pass over memory. */
int ir_linear_fwd_scaled(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                         int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                         void* stream);
/* The same with the kernel named by the caller (benchmarks, A/B, tests of every tile shape): */
#define IR_LIN_AUTO 0           /* what ir_linear_fwd / ir_linear_fwd_scaled pick (ir_linear_kernel_for) */
#define IR_LIN_X_STATIONARY 1
#define IR_LIN_TILED_FIRST 2    /* 2: 256x128, 3: 128x128, 4: 128x64, 5: 256x64, 6: 64x128, 7: 128x256, 8: 256x256 (rows x columns of Y per
                                   workgroup; 7 and 8 with 64 x 128 per wave), 9: 128x128 with the contraction split over two wave
                                   groups of the workgroup (K / 64 even; partial sums meet in LDS in a fixed order: deterministic, but
                                   not the same fp32 rounding as the single-pass tiles) */
int ir_linear_fwd_ex(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                     int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                     int32_t kernel, void* stream);
/* kernel id (IR_LIN_*) the automatic choice makes for a shape, or -1 when no kernel covers it */
int ir_linear_kernel_for(int64_t m, int32_t n, int32_t k, int32_t has_bias);

/*
 * ir_linear_fwd_stats - ir_linear_fwd_scaled that ALSO leaves the token statistics of output columns
 * [stats_col0, stats_col0 + stats_cols) behind: the AdaIN statistics pass (attn_processors.py:9-10, :244-245) fused into
 * the projection that produces V (ABI 6, round 4).
 *
 * The workgroup that has stored a (rows x 64-column) block of those columns re-reads it from its L2 and writes the block's
 * partial statistics - mean[64] | M2[64], fp32, of the ROUNDED 16-bit outputs, the values the attention kernel will read -
 * to stats_ws[(row_block * (stats_cols / 64) + head) * 128 ...], row blocks of ir_linear_stats_rows(m, n, k, has_bias)
 * rows in row order.  For the V third of a fused q/k/v projection (stats_col0 = 2C, stats_cols = C) over token sets of
 * L rows each (L % rows == 0) the partials of set s are blocks [s L / rows, (s + 1) L / rows): what
 * ir_adain_affine_from_partials / ir_token_stats_from_partials merge.  Same arithmetic as ir_adain_stats' own partial pass
 * (shifted single-pass sums, Chan merge); no pass over V, no extra launch.
 *   stats_col0, stats_cols, n: multiples of 64; m % rows == 0, rows = ir_linear_stats_rows(...) > 0 (0: this shape's kernel
 *   cannot emit them - use ir_adain_stats / ir_token_stats);  stats_ws: >= (m / rows) * (stats_cols / 64) * 512 bytes
 */
int ir_linear_stats_rows(int64_t m, int32_t n, int32_t k, int32_t has_bias);
int ir_linear_fwd_stats(int32_t dtype, int32_t x_is_f32, int64_t m, int32_t n, int32_t k, const void* x, int64_t x_ld, const void* w,
                        int64_t w_ld, const void* bias, void* y, int64_t y_ld, int32_t scale_cols, float col_scale,
                        int32_t stats_col0, int32_t stats_cols, float* stats_ws, size_t stats_ws_bytes, void* stream);
/*
 * ir_adain_affine_from_partials - the (a, b) of ir_adain_stats from those partials: one small launch per shared layer.
 *   style_ws: partials of V_self from the shared layer's own q/k/v projection (B sets of len_self rows, style_rows per block);
 *   content: EITHER content_ws / content_rows - partials of the reference V's from the K/V-capture layer's projection
 *   (B * N sets of len_ref rows, set index b * N + n) - OR content_mean / content_std, fp32 (B, N, H, 64), std without eps
 *   (ir_token_stats_from_partials, ir_token_stats: a per-identity cache); valid: optional int32 (B) on the device -
 *   references n >= valid[b] were zero-filled (ir_zero_invalid_refs) and count with statistics (0, 0).
 */
int ir_adain_affine_from_partials(int32_t batch, int32_t heads, int32_t n_refs, int32_t len_self, int32_t len_ref,
                                  const float* style_ws, int32_t style_rows, const float* content_ws, int32_t content_rows,
                                  const float* content_mean, const float* content_std, const int32_t* valid, float eps,
                                  float* a, float* b, void* stream);
/* mean and unbiased std (no eps) of n_sets matrices of `len` rows from their partials: fp32 (n_sets, H, 64) each */
int ir_token_stats_from_partials(int32_t n_sets, int32_t heads, int32_t len, const float* ws, int32_t rows, float* mean, float* std,
                                 void* stream);

/* library identity / diagnostics */
int ir_abi_version(void);                  /* == IR_ABI_VERSION */
const char* ir_build_info(void);           /* "gfx950 hipcc ... <date>" */
const char* ir_last_error_string(void);    /* thread-local, never NULL */

/*
 * ir_time_shared_attn_fwd - launch ir_shared_attn_fwd `iters` times on `stream` bracketed by
 * HIP events recorded on that same stream and return the average milliseconds per launch in
 * *ms_per_launch (bench.py's live roofline measurement; synchronises the stream).
 */
int ir_time_shared_attn_fwd(const ir_shared_attn_args* args, int32_t iters, void* stream, float* ms_per_launch);

/*
 * ir_bench_mfma_stream (ABI v7) - what the matrix pipe SUSTAINS on this device: `launches` back-to-back launches of an
 * MFMA-only stream of the attention kernels' instruction (v_mfma_f32_32x32x16 of `dtype`; two waves per SIMD on every CU,
 * iters x 16 MFMAs per wave, pseudo-random operands or - zero_operands != 0 - all zeros), bracketed by HIP events on
 * `stream`; returns the TFLOP/s of the bracketed launches in *tflops (synchronises the stream).  `scratch`: device memory,
 * >= ir_bench_mfma_stream_scratch_bytes().  Give it >= 0.1 s in all (e.g. iters 40000, launches 5) so that the board's
 * power controller settles: on random operands the 1400 W cap, not the 2.4 GHz of the datasheet peak, sets the result
 * (bench.py `roofline.at_power_cap`).  No reference counterpart: measurement only.
 */
size_t ir_bench_mfma_stream_scratch_bytes(void);
int ir_bench_mfma_stream(int32_t dtype, int32_t zero_operands, int32_t iters, int32_t launches, void* scratch, size_t scratch_bytes,
                         void* stream, float* tflops);

#ifdef __cplusplus
}
#endif
#endif /* INSTANTRESTORE_HIP_H */
