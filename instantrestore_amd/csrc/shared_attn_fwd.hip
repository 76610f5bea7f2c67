// shared_attn_fwd.hip - fused extended self-attention forward for gfx950 (MI355X).
//
// Computes, without ever materialising the probability matrix, what
// face_replace/models/attn_processors.py:232-264 computes with 3+2N head-split copies, N adain()
// calls, a torch.cat, baddbmm, softmax and bmm:
//
//     O = softmax(scale * Q [K_self? ; K_ref_0 ; ... ; K_ref_N-1]^T) [V_self? ; V'_ref_0 ; ...]
//     V'_ref_n = V_ref_n * a_n + b_n            (AdaIN as a per-channel affine)
//
// Design (CDNA4, 64-wide waves, head_dim 64):
//   * one workgroup = NW waves = NW*32 query rows of one (batch, head); each wave owns 32 rows;
//   * the K/V sequence is walked in place as a SEGMENT LIST [self?, ref0, ...] straight out of
//     the (B,L,C) / (B,N,L,C) activations - 64-key tiles, 128-byte head rows, coalesced 16-B
//     buffer loads (8 lanes per row; rows past a ragged segment end are zero-filled by the
//     descriptor's bounds check), register-staged into double-buffered LDS (32 KiB);
//   * both GEMMs are issued "swapped" on v_mfma_f32_32x32x16:  S^T = K Q^T  and  O^T = V^T P^T.
//     That puts a whole probability row (and its output row) in ONE lane pair (l, l^32): the row
//     max is an in-register v_max3 chain plus one v_permlane32_swap, the online-softmax rescale
//     is lane-local, and the exponentiated S registers ARE the B operand of the PV MFMA - no LDS
//     round trip, no permutes (the key order of the contraction is absorbed into the order V rows
//     are fetched);
//   * D = 64 makes this kernel VALU-issue bound, not MFMA bound (one exp + ~3 VALU per score
//     against 256 MFMA flops), so everything that can leave the VALU port does:
//       - row sums ride on the matrix pipe: a constant "ones" A-fragment appended as a 65th
//         V^T row gives l = P 1 from 4 extra MFMAs per tile instead of 32 v_add per lane;
//       - the AdaIN affine is NOT applied to V tiles; it is folded algebraically per segment:
//         O += (P_seg V_seg) o a_seg + rowsum(P_seg) b_seg  at each segment boundary (two FMAs
//         per accumulator per SEGMENT instead of per-tile unpack/FMA/pack work), so the
//         renormalised V is never formed anywhere;
//       - scale-and-subtract runs as packed v_pk_fma_f32; maxima as v_max3;
//   * K tile: XOR-swizzled 16-B slots -> conflict-free ds_read_b128 A-operand fetches;
//     V tile: row-major with a 64-B half swap -> conflict-free ds_read_b64_tr_b16 transposed
//     fetches of the V^T A-operand (SQ_LDS_BANK_CONFLICT = 0 measured);
//   * softmax in fp32 in the exp2 domain; accumulators are rescaled only when a running max
//     actually moved (exact), otherwise the multiply pass is skipped;
//   * blockIdx is remapped so the query blocks that share one (b,h)'s K/V sit on one XCD's L2.
#include "ir_common.h"
#include "ir_kernels.h"

#include <stdlib.h>

#ifndef IR_W128_DEFAULT
#define IR_W128_DEFAULT 0   // 1: the 128-row kernel is the default wherever the 64-row kernel was (set by build.sh once measured faster)
#endif

#ifdef IR_ABLATIONS   // the first, straight-line kernel (variants 1/2) and its timing ablations: development builds only
namespace {

constexpr int KVB = IR_KV_TILE;            // 64 keys per tile
constexpr int TILE_BYTES = KVB * 64 * 2;   // 8 KiB per K (or V) tile

// ABL: ablation bits for timing experiments only (results are WRONG when non-zero):
//   1 = no staging (no global loads / LDS writes / barriers), 2 = no softmax max/exp,
//   4 = no LDS fragment reads (constant operands)
template <typename T, int NW, bool FOLD, int ABL = 0>
__global__ void __launch_bounds__(NW * 64, 2) shared_attn_fwd_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NT = NW * 64;          // threads
  constexpr int QB = NW * 32;          // query rows per workgroup
  constexpr int CH = (KVB * 8) / NT;   // 16-B chunks per thread per matrix per tile
  static_assert(CH >= 1, "too many threads for a 64x64 tile");

  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;   // which 32-lane half
  const int lq = lane & 31;   // query row within the wave / MFMA column

  // ---- which (batch, head, query block) ---------------------------------------------------
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = lin / p.nqb;
  const int qb = lin - bh * p.nqb;
  const int b = bh / p.H;
  const int h = bh - b * p.H;

  // ---- Q fragments: B operand of S^T = K Q^T; lane holds Q[row lq][d = 16ks + 8hi .. +7] -----
  const int qrow = qb * QB + wid * 32 + lq;
  const int qrow_c = qrow < p.Lq ? qrow : p.Lq - 1;
  v8 qf[4];
  {
    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)qrow_c * p.q_sl + (int64_t)h * p.q_sh + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8*)(qp + ks * 16);
  }
  // "ones" A-fragment: row 0 of a virtual third 32-row block of V^T is all ones, rows 1..31 zero
  v8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (lq == 0) ? (T)1.0f : (T)0.0f;

  // ---- staging coordinates (global -> registers -> LDS) ------------------------------------
  const int slot = tid & 7;  // which 16-B (8 element) slot of the 128-B head row
  int srow[CH], koff[CH], voff[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int row = (tid >> 3) + c * (NT / 8);
    srow[c] = row;
    koff[c] = row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
    voff[c] = row * 128 + ((slot ^ (((row >> 1) & 1) << 2)) << 4);
  }

  // ---- LDS read offsets ---------------------------------------------------------------------
  int kread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) kread[ks] = lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);
  int vread[2];
  {
    const int m = lane & 15, g = (lane >> 4) & 1;
    const int sw = (m >> 3) & 1;
#pragma unroll
    for (int db = 0; db < 2; ++db)
      vread[db] = (4 * hi + (m >> 2)) * 128 + ((db ^ sw) << 6) + 32 * g + 8 * (m & 3);
  }

  // ---- segment iterator of the prefetch stream (all wave-uniform) --------------------------
  const int nseg = p.include_self + p.N;
  __amdgpu_buffer_rsrc_t krs, vrs;
  int kstep = 0, vstep = 0;  // bytes per 64-key tile
  int sntile = 0;
  unsigned kvo[CH], vvo[CH];  // per-thread byte offsets of this thread's chunks inside the segment

  auto seg_setup = [&](int s) {
    const T* sk;
    const T* sv;
    int ksl_b, vsl_b, slen;
    if (p.include_self && s == 0) {
      sk = (const T*)p.k_self + (int64_t)b * p.ks_sb + (int64_t)h * p.ks_sh;
      sv = (const T*)p.v_self + (int64_t)b * p.vs_sb + (int64_t)h * p.vs_sh;
      ksl_b = (int)p.ks_sl * 2; vsl_b = (int)p.vs_sl * 2; slen = p.Ls; sntile = p.tiles_self;
    } else {
      const int n = s - p.include_self;
      sk = (const T*)p.k_ref + (int64_t)b * p.kr_sb + (int64_t)n * p.kr_sn + (int64_t)h * p.kr_sh;
      sv = (const T*)p.v_ref + (int64_t)b * p.vr_sb + (int64_t)n * p.vr_sn + (int64_t)h * p.vr_sh;
      ksl_b = (int)p.kr_sl * 2; vsl_b = (int)p.vr_sl * 2; slen = p.Lr; sntile = p.tiles_ref;
    }
    // num_records ends right after the last valid head row: later rows read as zeros
    krs = __builtin_amdgcn_make_buffer_rsrc((void*)sk, 0, (slen - 1) * ksl_b + 128, 0x00020000);
    vrs = __builtin_amdgcn_make_buffer_rsrc((void*)sv, 0, (slen - 1) * vsl_b + 128, 0x00020000);
    kstep = KVB * ksl_b;
    vstep = KVB * vsl_b;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      kvo[c] = (unsigned)(srow[c] * ksl_b + slot * 16);
      vvo[c] = (unsigned)(srow[c] * vsl_b + slot * 16);
    }
  };

  u32x4 kreg[CH], vreg[CH];
  int seg = 0, t0 = 0;

  auto issue_loads = [&]() {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      kreg[c] = __builtin_amdgcn_raw_buffer_load_b128(krs, kvo[c], 0, 0);
      vreg[c] = __builtin_amdgcn_raw_buffer_load_b128(vrs, vvo[c], 0, 0);
      kvo[c] += kstep;
      vvo[c] += vstep;
    }
  };
  auto stage_write = [&](int buf) {
    unsigned char* Kb = smem + buf * (2 * TILE_BYTES);
    unsigned char* Vb = Kb + TILE_BYTES;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      *(u32x4*)(Kb + koff[c]) = kreg[c];
      *(u32x4*)(Vb + voff[c]) = vreg[c];
    }
  };
  auto advance = [&]() {
    if (++t0 == sntile) {
      t0 = 0;
      if (++seg < nseg) seg_setup(seg);
    }
  };

  // ---- accumulators ----------------------------------------------------------------------
  // o0/o1: O^T of the CURRENT segment, rows d = 32*db + crow(r,hi), column = query lq
  // o2   : ones block; o2[0] of the hi=0 lane = row sum of the current segment
  // ot0/ot1/l_tot (FOLD only): everything folded so far, AdaIN affine applied
  f32x16 o0, o1, o2, ot0, ot1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; o2[r] = 0.f; ot0[r] = 0.f; ot1[r] = 0.f; }
  float l_tot = 0.f;
  float m_run = -INFINITY;  // running max of the RAW (unscaled) scores of this lane pair's row
  const float c2 = p.scale_log2;

  // compute-stream position (lags the prefetch stream by one tile)
  int cseg = 0, ct0 = 0;
  int c_ntile = (p.include_self ? p.tiles_self : p.tiles_ref);
  int c_len = (p.include_self ? p.Ls : p.Lr);

  // ---- prologue ------------------------------------------------------------------------
  seg_setup(0);
  issue_loads();
  stage_write(0);
  advance();
  // Retire the Q-fragment loads HERE.  hipcc's waitcnt pass otherwise carries them as "maybe
  // outstanding" into the loop and guards the first use of qf[] in every iteration with a
  // vmcnt(1)/vmcnt(0) that also drains the K/V prefetch issued a few instructions earlier - a
  // full memory latency at the top of every tile (found in the .s; the prefetch was dead).
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));
  __syncthreads();

  for (int ti = 0; ti < p.ntiles; ++ti) {
    const int buf = ti & 1;
    const bool has_next = (ti + 1 < p.ntiles);
    if (has_next && !(ABL & 1)) issue_loads();  // HBM/L2 latency hides under this tile's math

    const unsigned char* Kb = smem + buf * (2 * TILE_BYTES);
    const unsigned char* Vb = Kb + TILE_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks x four 16-wide d steps ---------------------------
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const v8 a0 = (ABL & 4) ? qf[(ks + 1) & 3] : *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + kread[ks]);
      const v8 a1 = (ABL & 4) ? qf[(ks + 2) & 3] : *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + 32 * 128 + kread[ks]);
      s0 = Tr::mfma(a0, qf[ks], s0);
      s1 = Tr::mfma(a1, qf[ks], s1);
    }
    // lane (lq,hi) now holds, for query lq, keys crow(r,hi) = (r&3) + 8*(r>>2) + 4*hi (+32 for s1)

    const int valid_cur = c_len - ct0 * KVB;
    if (valid_cur < KVB) {  // ragged last tile of a segment (wave-uniform branch)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= valid_cur) s0[r] = -INFINITY;
        if (key + 32 >= valid_cur) s1[r] = -INFINITY;
      }
    }

    // ---- online softmax (exp2 domain) -----------------------------------------------------
    // The v_max3 chain below is inline asm reading MFMA results: hipcc pads MFMA->VALU hazards
    // only for instructions it can see, so the 12 wait states an 8-pass MFMA result needs are
    // spelled out here, tied to both accumulators so nothing that reads them moves above it.
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(s0), "+v"(s1));
    if (!(ABL & 2)) {
    // two independent v_max3 chains (one per key block), then one cross-half exchange:
    // permlane32_swap(x, x) leaves {own, partner} (in either order) in the two results
    float mxa = max3(s0[0], s0[1], s0[2]);
    float mxb = max3(s1[0], s1[1], s1[2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) {
      mxa = max3(mxa, s0[r], s0[r + 1]);
      mxb = max3(mxb, s1[r], s1[r + 1]);
    }
    float mx = max3(mxa, mxb, max3(s0[15], s1[15], s1[15]));
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), mx);
    }
    const float m_new = fmaxf(m_run, mx);
    const float mc = m_new * c2;
    if (__any(m_new != m_run)) {  // some row's max moved: rescale (exact; skipped otherwise)
      const float alpha = fast_exp2(m_run * c2 - mc);
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      o2[0] *= alpha;
      if (FOLD) {
        l_tot *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ot0[r] *= alpha; ot1[r] *= alpha; }
      }
      m_run = m_new;
    }
    {
      const f32x2 cc = {c2, c2};
      const f32x2 nm = {-mc, -mc};
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 t0v = {s0[r], s0[r + 1]};
        f32x2 t1v = {s1[r], s1[r + 1]};
        t0v = __builtin_elementwise_fma(t0v, cc, nm);  // v_pk_fma_f32
        t1v = __builtin_elementwise_fma(t1v, cc, nm);
        s0[r] = fast_exp2(t0v[0]); s0[r + 1] = fast_exp2(t0v[1]);
        s1[r] = fast_exp2(t1v[0]); s1[r + 1] = fast_exp2(t1v[1]);
      }
    }
    }  // !(ABL & 2)

    // P^T fragments (B operand of O^T = V^T P^T): registers 8ks..8ks+7 of key block kb
    v8 pk[2][2];
    {
      const f32x8 p00 = __builtin_shufflevector(s0, s0, 0, 1, 2, 3, 4, 5, 6, 7);
      const f32x8 p01 = __builtin_shufflevector(s0, s0, 8, 9, 10, 11, 12, 13, 14, 15);
      const f32x8 p10 = __builtin_shufflevector(s1, s1, 0, 1, 2, 3, 4, 5, 6, 7);
      const f32x8 p11 = __builtin_shufflevector(s1, s1, 8, 9, 10, 11, 12, 13, 14, 15);
      pk[0][0] = __builtin_convertvector(p00, v8);
      pk[0][1] = __builtin_convertvector(p01, v8);
      pk[1][0] = __builtin_convertvector(p10, v8);
      pk[1][1] = __builtin_convertvector(p11, v8);
    }

    // ---- O^T += V^T P^T : A operand = V^T fetched with the LDS transpose read ----------------
    // k index 8*hi + i of step (kb,ks) is key 32kb + 16ks + 8(i>>2) + 4hi + (i&3): exactly the
    // key order the P registers already have.  The third MFMA of every step multiplies the
    // constant ones fragment: its row 0 accumulates the row sum of the (rounded) P.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = (32 * kb + 16 * ks) * 128;
        o2 = Tr::mfma(ones, pk[kb][ks], o2);
        if (ABL & 4) {
          o0 = Tr::mfma(qf[kb + ks], pk[kb][ks], o0);
          o1 = Tr::mfma(qf[3 - kb - ks], pk[kb][ks], o1);
        } else {
          const s16x4 a00 = lds_read_tr16(Vb + vread[0] + off);
          const s16x4 a01 = lds_read_tr16(Vb + vread[0] + off + 8 * 128);
          const s16x4 a10 = lds_read_tr16(Vb + vread[1] + off);
          const s16x4 a11 = lds_read_tr16(Vb + vread[1] + off + 8 * 128);
          o0 = Tr::mfma(join_tr<v8>(a00, a01), pk[kb][ks], o0);
          o1 = Tr::mfma(join_tr<v8>(a10, a11), pk[kb][ks], o1);
        }
      }
    }

    // ---- segment boundary of the compute stream: fold the AdaIN affine ----------------------
    if (++ct0 == c_ntile) {
      if (FOLD) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o2[0]), __float_as_uint(o2[0]), false, false);
        const float lseg = __uint_as_float(sw[0]);  // row sum from the hi=0 lane, in both halves
        l_tot += lseg;
        const bool is_ref = !(p.include_self && cseg == 0);
        if (is_ref && p.aa != nullptr) {
          const int n = cseg - p.include_self;
          const int64_t ao = ((int64_t)(b * p.N + n) * p.H + h) * 64 + 4 * hi;
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 a0 = *(const f32x4*)(p.aa + ao + 8 * g4), a1 = *(const f32x4*)(p.aa + ao + 32 + 8 * g4);
            const f32x4 b0 = *(const f32x4*)(p.ab + ao + 8 * g4), b1 = *(const f32x4*)(p.ab + ao + 32 + 8 * g4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              ot0[4 * g4 + i] = __builtin_fmaf(o0[4 * g4 + i], a0[i], __builtin_fmaf(lseg, b0[i], ot0[4 * g4 + i]));
              ot1[4 * g4 + i] = __builtin_fmaf(o1[4 * g4 + i], a1[i], __builtin_fmaf(lseg, b1[i], ot1[4 * g4 + i]));
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) { ot0[r] += o0[r]; ot1[r] += o1[r]; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
        o2[0] = 0.f;
      }
      ct0 = 0;
      ++cseg;
      c_ntile = p.tiles_ref;
      c_len = p.Lr;
    }

    if (!(ABL & 1)) {
      if (has_next) {
        stage_write(buf ^ 1);
        advance();
      }
      __syncthreads();
    }
  }

  // ---- epilogue: normalise, store O (and LSE) ----------------------------------------------
  float l_fin;
  if (FOLD) {
    l_fin = l_tot;
  } else {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o2[0]), __float_as_uint(o2[0]), false, false);
    l_fin = __uint_as_float(sw[0]);
  }
  const float inv = 1.0f / l_fin;
  if (qrow < p.Lq) {
    T* op = (T*)p.out + (int64_t)b * p.o_sb + (int64_t)qrow * p.o_sl + (int64_t)h * p.o_sh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 x0, x1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x0[i] = (FOLD ? ot0[4 * g4 + i] : o0[4 * g4 + i]) * inv;
        x1[i] = (FOLD ? ot1[4 * g4 + i] : o1[4 * g4 + i]) * inv;
      }
      *(v4*)(op + 8 * g4 + 4 * hi) = __builtin_convertvector(x0, v4);
      *(v4*)(op + 32 + 8 * g4 + 4 * hi) = __builtin_convertvector(x1, v4);
    }
    if (p.lse != nullptr && hi == 0)
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_run * p.scale + __logf(l_fin);
  }
}

template <typename T, int NW, bool FOLD, int ABL = 0>
hipError_t launch(const AttnKParams& p0, hipStream_t s) {
  AttnKParams p = p0;
  constexpr int QB = NW * 32;
  p.nqb = (p.Lq + QB - 1) / QB;
  const int grid = p.B * p.H * p.nqb;
  hipLaunchKernelGGL((shared_attn_fwd_kernel<T, NW, FOLD, ABL>), dim3(grid), dim3(NW * 64), 0, s, p);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_t(const AttnKParams& p, int nw, hipStream_t s) {
  const bool fold = (p.aa != nullptr);
  if (nw == 8) return fold ? launch<T, 8, true>(p, s) : launch<T, 8, false>(p, s);
  return fold ? launch<T, 4, true>(p, s) : launch<T, 4, false>(p, s);
}

}  // namespace
#endif  // IR_ABLATIONS

// variant & 31 selects the kernel (per-call tuning field of ir_shared_attn_args; 0 = default dispatch, see
// ir_attn_default_is_w64).  Product library:
//   16 128 query rows per wave, ONE wave per SIMD, hand-placed instruction stream (shared_attn_fwd_w128.hip; pre-scaled Q only)
//   13 64 query rows per wave, 8-wave (512-row) workgroups (shared_attn_fwd_w64.hip)     12 the same, 4 waves
//   10 software-pipelined 32-row kernel, 4 waves, asm-issued LDS-DMA staging, lazy max (shared_attn_fwd_pipe.hip)
//    7 the same with an exact (every-change) rescale
//   11 10 + pre-scaled Q, reference through the MFMA C operand (opt-in fast mode: one more rounding of Q)
//   14 10 with the next tile's QK^T issued before the row max
//   18 11 with the reference checked after the exponentials (no row max on ordinary tiles; K ring of 3)
// Development builds only (tools/experiments/build.sh, -DIR_ABLATIONS; documented experiments, NOTES.md): 1/2 this file's
// straight-line kernel with 8 / 4 waves, 3/4 pipelined with register staging, 6 pipelined + builtin LDS-DMA, 9 straight
// schedule at 3 waves/SIMD; 20-28 the energy / timing ablations of the 64-row QS kernel (shared_attn_fwd_w64.hip, WRONG
// results); variant >> 5: ablation bits of the 32-row kernels (timing experiments, WRONG results).  Those variants write
// 16-bit results only: they are rejected together with IR_FLAG_OUT_F32.  (Rounds 1-3 also carried 8 ping-pong wave groups,
// 15 the 64-row kernel with rotated phases, 16 one wave per SIMD, 17 a three-stage 32-row pipeline: measured negative
// results - NOTES.md 4.1b, 4.1b', profiles/r1_pp_phase_trace.txt, r2_sp_ablation.txt - whose sources were second copies of
// product kernel bodies and left the tree in round 5.)
bool ir_attn_variant_available(int variant) {
  const int base = variant & 31;
#ifdef IR_ABLATIONS
  return (base <= 18 && base != 8 && base != 15 && base != 17) || (base >= 20 && base < 20 + ir_w64_abl_count());
#else
  if ((variant >> 5) != 0) return false;
  return base == 0 || base == 7 || (base >= 10 && base <= 14) || base == 16 || base == 18;   // 16 (SP64) and 17 (TP32): development builds, like 1-6, 8, 9, 15
#endif
}

// Default dispatch: the 64-rows-per-wave kernel in 512-row workgroups for query axes of >= 4096 rows whose (b, h, 512-row)
// items fill the chip or whose K/V walk is long; the 32-row pipelined kernel below that
bool ir_attn_default_is_w64(const AttnKParams& p) {
  const long items512 = (long)p.B * p.H * ((p.Lq + 511) / 512);
  return p.Lq >= 4096 && (items512 >= 256 || p.ntiles >= 128);
}

// Round 6: where the 64-row kernel would run AND the call is in the 128-row kernel's domain (pre-scaled Q, whole tiles, no
// valid_refs / seg_mass) the one-wave-per-SIMD kernel takes it.  IR_ATTN_W128=0 / 1 overrides the rule for A/B runs.
bool ir_attn_default_is_w128(const AttnKParams& p) {
  static const int env = [] { const char* e = getenv("IR_ATTN_W128"); return e == nullptr ? -1 : (e[0] == '0' ? 0 : 1); }();
  if (env == 0) return false;
  if (!ir_attn_w128_supports(p)) return false;
  if (env == 1) return p.Lq >= 1024;
  if (IR_W128_DEFAULT == 0) return false;
  if (ir_attn_default_is_w64(p)) return true;
  // 32x32-token class (1024 <= Lq < 4096; cfg 4's and cfg 5's shapes - at cfg 2's own the two kernels measure equal and the 32-row
  // kernel stays): the 512-row work items pay off once the K/V walk is long or the item grid is several rounds deep
  // (profiles/r6_layer_classes.txt: cfg 5 shared +22 %, cfg 4 shared +10 %, capture forms of 1280+ items +3...5 %; 640 items of
  // 16 tiles -10 %)
  if (p.Lq < 1024) return false;
  const long items512 = (long)p.B * p.H * ((p.Lq + 511) / 512);
  return p.ntiles >= 64 ? (items512 >= 512 || p.ntiles >= 128) : items512 >= 1280;
}

// ABI v9 (seg_mass): the forward kernels left, per row, the cumulative log-sum-exp c_0 <= c_1 <= ... <= c_{S-1} (= the row's LSE)
// through each K/V segment; the mass of segment s is exp(c_s - c_{S-1}) - exp(c_{s-1} - c_{S-1}).  In place, one thread per row;
// the masses of a row telescope to exactly 1.
__global__ void __launch_bounds__(256) seg_mass_finish_kernel(float* __restrict__ cum, long rows, int S) {
  const long r = (long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float* c = cum + r * S;
  const float tot = c[S - 1];
  float prev = 0.f;
  for (int s = 0; s < S; ++s) {
    const float e = s == S - 1 ? 1.f : __expf(c[s] - tot);
    c[s] = e - prev;
    prev = e;
  }
}

hipError_t ir_launch_seg_mass_finish(const AttnKParams& p, hipStream_t s) {
  const long rows = (long)p.B * p.H * p.Lq;
  if (rows == 0 || p.nseg_out <= 0) return hipSuccess;
  hipLaunchKernelGGL(seg_mass_finish_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, p.seg_cum, rows, p.nseg_out);
  return hipGetLastError();
}

static hipError_t launch_attn_kernel(const AttnKParams& p, int dtype, int variant, hipStream_t s);

hipError_t ir_launch_shared_attn_fwd(const AttnKParams& p, int dtype, int variant, hipStream_t s) {
  if (!ir_attn_variant_available(variant)) return hipErrorInvalidValue;
  if (p.seg_cum != nullptr) {
    // by-product of the 64-row and the pipelined 32-row kernels (every product kernel); the development-only experiments never learned it
    // (the 8-wave 64-row kernel and the 32-row kernel's two default forms carry the MASS instantiation: tuning 0, 11, 13, 14)
    const int b0 = variant & 31;
    if ((variant >> 5) != 0 || !(b0 == 0 || b0 == 11 || b0 == 13 || b0 == 14)) return hipErrorInvalidValue;   // (16 has no MASS form)
  }
  const hipError_t e = launch_attn_kernel(p, dtype, variant, s);
  if (e != hipSuccess || p.seg_cum == nullptr) return e;
  return ir_launch_seg_mass_finish(p, s);
}

static hipError_t launch_attn_kernel(const AttnKParams& p, int dtype, int variant, hipStream_t s) {
#ifdef IR_ABLATIONS
  {   // the experiments that never learned IR_FLAG_OUT_F32 would write 16-bit data into an fp32 buffer
    const int b0 = variant & 31;
    if (p.out_f32 && (b0 == 1 || b0 == 2 || b0 >= 20 || (variant >> 5) != 0)) return hipErrorInvalidValue;
  }
  const int abl = variant >> 5;
  if (abl != 0 && (variant & 31) == 3) return ir_launch_shared_attn_fwd_pipe_abl(p, abl, s);
  if (abl != 0) {
    switch (abl & 7) {
      case 1: return launch<__bf16, 4, false, 1>(p, s);
      case 2: return launch<__bf16, 4, false, 2>(p, s);
      case 3: return launch<__bf16, 4, false, 3>(p, s);
      case 4: return launch<__bf16, 4, false, 4>(p, s);
      case 5: return launch<__bf16, 4, false, 5>(p, s);
      case 6: return launch<__bf16, 4, false, 6>(p, s);
      default: return launch<__bf16, 4, false, 7>(p, s);
    }
  }
#endif
  const int base = variant & 31;
  // pre-scaled Q: the 64-row kernel's QS instantiation where the default rule (or IR_TUNE_W64X8) takes that kernel, the
  // 32-row kernel's reference-through-C form for every other shape.  fp32 output (IR_FLAG_OUT_F32): every product kernel
  // stores its result before the rounding when asked - what the parity tests look at is the kernel that ships
#ifdef IR_ABLATIONS
  if (base >= 20) return ir_launch_shared_attn_fwd_w64_abl(p, dtype, base - 20, s);
#endif
  if (base == 16) return ir_launch_shared_attn_fwd_w128(p, dtype, s);   // (refuses what it does not cover)
  if (p.q_prescaled) {
    if (base == 0 && ir_attn_default_is_w128(p)) return ir_launch_shared_attn_fwd_w128(p, dtype, s);
    if ((base == 0 && ir_attn_default_is_w64(p)) || base == 13) return ir_launch_shared_attn_fwd_w64x8(p, dtype, s);
    // the 32-row kernel's pre-scaled-Q form (no Q rounding of its own), with the row max of every tile.  Its
    // check-after-the-exponentials form (IR_TUNE_PIPE32_POSTCHECK, round 3) is parity-green and measures the SAME time
    // on both short layer classes (profiles/r3_layer_classes_cfg2_presc.txt: this kernel is bound by its LDS fragment
    // reads, not by vector instructions), at 8 KiB more LDS - so it stays opt-in
    return ir_launch_shared_attn_fwd_pipe(p, dtype, base == 18 ? 18 : 11, s);
  }
  switch (base) {
    case 0:
      if (ir_attn_default_is_w64(p)) return ir_launch_shared_attn_fwd_w64x8(p, dtype, s);
      return ir_launch_shared_attn_fwd_pipe(p, dtype, 14, s);
    case 7: case 10: case 11: case 14: case 18: return ir_launch_shared_attn_fwd_pipe(p, dtype, base, s);
    case 12: return ir_launch_shared_attn_fwd_w64(p, dtype, s);
    case 13: return ir_launch_shared_attn_fwd_w64x8(p, dtype, s);
#ifdef IR_ABLATIONS
    case 3: return ir_launch_shared_attn_fwd_pipe(p, dtype, 4, s);   // register staging, 4 waves
    case 4: return ir_launch_shared_attn_fwd_pipe(p, dtype, 8, s);   // register staging, 8 waves
    case 6: case 9: return ir_launch_shared_attn_fwd_pipe(p, dtype, base, s);
    case 1: case 2: {
      const int nw = (base == 1) ? 8 : 4;
      return dtype == 1 ? launch_t<__bf16>(p, nw, s) : launch_t<_Float16>(p, nw, s);
    }
#endif
    default: return hipErrorInvalidValue;
  }
}
