// shared_attn_fwd_sp.hip - software-pipelined, ONE WAVE PER SIMD form of the fused extended self-attention
// forward (gfx950).  Same math, layouts and C-ABI contract as the other kernels (shared_attn_fwd.hip has the
// derivation: swapped contractions S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_32x32x16, a probability row in
// one lane pair, K/V walked in place as a segment list, AdaIN folded per segment in a ratio frame).
//
// Why another schedule.  At head_dim 64 a 64-key tile costs a wave 32 MFMAs (1024 matrix-pipe cycles) and ~260
// single-issue VALU instructions (max3, fma, exp, add, cvt_pk per score).  In the 64-row kernel
// (shared_attn_fwd_w64.hip) a wave runs QK^T -> softmax -> PV back to back and its two waves per SIMD sit in the
// same phase between two barriers: matrix and vector time add up (78 cycles per MFMA, matrix pipe 40 % busy).
// tools/ubench/gapfill.hip (profiles/r2_ubench_gapfill.txt) measured what ONE wave can hide: between two MFMAs on
// DIFFERENT accumulators ~4.5 single-issue VALU instructions are free and each further one costs ~5.5 cycles -
// 8 fillers per gap run at 52.6 cycles per MFMA.  So the wave itself must carry both kinds of work at once:
//   * one wave per SIMD (4 waves = 256 query rows per workgroup, the whole 512-register file per wave), each
//     wave two 32-row blocks A and B, both with a DOUBLE-BUFFERED score block, B running half a tile behind A:
//         region X(i):  MFMA  PV_B(i-1), QK_A(i+1)     VALU  exp/sum/convert of S_A(i),  row max of S_B(i)
//         region Y(i):  MFMA  QK_B(i+1), PV_A(i)       VALU  exp/sum/convert of S_B(i),  row max of S_A(i+1)
//     so every MFMA gap has independent vector work of the OTHER row block / tile to carry;
//   * the interleave is spelled out, not left to the scheduler: each region is 16 clusters of
//     { 1 MFMA, ~8 VALU, <= 1 LDS read } separated by sched_barrier(0); the four accumulators of a region rotate
//     (same accumulator every 4th MFMA, the arrangement the microbenchmark measured);
//   * K fragments of a tile are read from LDS once and serve both row blocks (held in registers across the two
//     regions), V^T fragments likewise - 24 LDS reads per 32 MFMAs as in the 64-row kernel;
//   * rescale decisions (lazy max), ragged-tile masks and the AdaIN segment boundary sit BETWEEN regions, after
//     the pending P V of that row block has completed (the safe order for a deferred rescale);
//   * K ring of 3, V ring of 2 tiles in LDS, filled by asm-issued LDS-DMA a full iteration ahead; one barrier
//     per tile.
// PRESC (IR_FLAG_Q_PRESCALED): Q arrives as Q*scale*log2(e), the running reference enters through the C operand
// of the first QK^T MFMA of a tile, the scores leave the matrix pipe as exponents: no multiply-add per score.
#ifdef IR_ABLATIONS   // documented experiment (variant 16): development builds only

#include <type_traits>

#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int KVB = IR_KV_TILE;
constexpr int TILE_BYTES = KVB * 64 * 2;  // 8 KiB
constexpr int KRING = 3, VRING = 2;
// SP_ABL: timing ablations (development only, WRONG results): 1 no barrier, 2 no DMA, 4 no LDS fragment reads,
// 8 no boundary work (max exchange / rescale), 16 no vector work, 32 no MFMAs, 64 no s_nop pads
#ifndef SP_ABL
#define SP_ABL 0
#endif
#ifndef SP_PV_MODE
#define SP_PV_MODE 2   // 2: asm MFMA, O accumulators in VGPRs (0: AGPRs - gives wrong results, kept for the record; 1: builtin)
#endif

template <typename T>
struct Pair2;
template <>
struct Pair2<__bf16> { typedef __attribute__((ext_vector_type(2))) __bf16 type; };
template <>
struct Pair2<_Float16> { typedef __attribute__((ext_vector_type(2))) _Float16 type; };


// One MFMA gap, spelled out: the MFMA and the single-issue vector work that rides in its shadow as two adjacent
// volatile asm statements (the compiler's schedulers reorder builtin MFMAs and pure VALU code at will - neither
// sched_barrier clusters nor IGroupLP pipelines came out as written - while volatile asm statements keep their
// order; register allocation, LDS waits and everything between the regions stay the compiler's).  Vector work of a
// cluster = two scores of the tile being exponentiated (fma, exp, row-sum add, convert to a 16-bit pair) and one
// v_max3 of the tile being reduced; exp -> add are two instructions apart (trans-use hazard), the max sits between.
// Register files: S blocks in VGPRs (VALU reads them), O blocks in AGPRs (touched by VALU only on the rare rescale /
// fold paths), K / V / Q fragments in AGPRs (MFMA operands only), probabilities in VGPRs.
// ACC_A: accumulator in AGPRs;  CMODE: 0 accumulate, 1 C = 0, 2 C = `cblk` (PRESC: the negated reference);
// B_A: B operand in AGPRs (Q fragments) or VGPRs (probabilities).
template <typename T, bool ACC_A, int CMODE, bool B_A, typename V8>
__device__ __forceinline__ void sp_mfma(f32x16& acc, const V8& a, const V8& b, const f32x16& cblk) {
  constexpr bool BF = std::is_same<T, __bf16>::value;
  if (SP_ABL & 32) { asm volatile("" : "+v"(acc)); return; }
#define IR_SP_M2(OPS, OUTC, BC, EXTRA)                                                                   \
  if (BF) asm volatile("v_mfma_f32_32x32x16_bf16 " OPS : OUTC(acc) : "a"(a), BC(b) EXTRA);               \
  else asm volatile("v_mfma_f32_32x32x16_f16 " OPS : OUTC(acc) : "a"(a), BC(b) EXTRA);
  if (CMODE == 0) {
    if (ACC_A) { if (B_A) { IR_SP_M2("%0, %1, %2, %0", "+a", "a", ) } else { IR_SP_M2("%0, %1, %2, %0", "+a", "v", ) } }
    else { if (B_A) { IR_SP_M2("%0, %1, %2, %0", "+v", "a", ) } else { IR_SP_M2("%0, %1, %2, %0", "+v", "v", ) } }
  } else if (CMODE == 1) {
    if (B_A) { IR_SP_M2("%0, %1, %2, 0", "=&v", "a", ) } else { IR_SP_M2("%0, %1, %2, 0", "=&v", "v", ) }
  } else {
#define IR_SP_CB , "v"(cblk)
    if (B_A) { IR_SP_M2("%0, %1, %2, %3", "=&v", "a", IR_SP_CB) } else { IR_SP_M2("%0, %1, %2, %3", "=&v", "v", IR_SP_CB) }
#undef IR_SP_CB
  }
#undef IR_SP_M2
}

// `hold`: the VGPR B operand (probabilities) of the MFMA issued just before.  To the compiler those registers are dead
// once the MFMA statement has been emitted and it hands them to this block's temporaries - but the matrix pipe is still
// reading them (found the hard way: O of one row block came out as garbage whenever the allocator made that choice).
// Listing them as an (unused) input keeps them untouched until this block is over, > 32 cycles after the MFMA issued.
template <typename T, bool PRESC, typename V8>
__device__ __forceinline__ void sp_vec(float s0, float s1, float c2, float nmc, float& l0, float& l1, unsigned& pk,
                                       float& mx, float m0, float m1, const V8& hold) {
  if (SP_ABL & 16) { asm volatile("" : "+v"(l0), "+v"(l1), "=v"(pk), "+v"(mx) : "v"(s0), "v"(s1)); return; }
  float t0, t1;   // separate results: writing the exponentials over the score registers costs a copy per cluster
  constexpr bool BF = std::is_same<T, __bf16>::value;
#define IR_SP_REST(CVT, X0, X1)                         \
  "v_exp_f32 %[t0], " X0 "\n\t"                         \
  "v_exp_f32 %[t1], " X1 "\n\t"                         \
  "v_max3_f32 %[mx], %[mx], %[m0], %[m1]\n\t"           \
  "v_add_f32 %[l0], %[l0], %[t0]\n\t"                   \
  "v_add_f32 %[l1], %[l1], %[t1]\n\t"                   \
  CVT " %[pk], %[t0], %[t1]"
#define IR_SP_FMA                                       \
  "v_fma_f32 %[t0], %[s0], %[c2], %[nmc]\n\t"           \
  "v_fma_f32 %[t1], %[s1], %[c2], %[nmc]\n\t"
#define IR_SP_OUT [t0] "=&v"(t0), [t1] "=&v"(t1), [l0] "+v"(l0), [l1] "+v"(l1), [pk] "=v"(pk), [mx] "+v"(mx)
#define IR_SP_IN [s0] "v"(s0), [s1] "v"(s1), [m0] "v"(m0), [m1] "v"(m1), [hold] "v"(hold)
  if (SP_ABL & 128) {   // timing ablation: five vector instructions per gap (no multiply-add, no row max) - the pre-scaled, check-after form
    asm volatile("v_exp_f32 %[t0], %[s0]\n\tv_exp_f32 %[t1], %[s1]\n\tv_add_f32 %[l0], %[l0], %[t0]\n\tv_add_f32 %[l1], %[l1], %[t1]\n\tv_cvt_pk_bf16_f32 %[pk], %[t0], %[t1]"
                 : IR_SP_OUT : IR_SP_IN);
  } else if (PRESC) {
    if (BF) asm volatile(IR_SP_REST("v_cvt_pk_bf16_f32", "%[s0]", "%[s1]") : IR_SP_OUT : IR_SP_IN);
    else asm volatile(IR_SP_REST("v_cvt_pk_f16_f32", "%[s0]", "%[s1]") : IR_SP_OUT : IR_SP_IN);
  } else {
    if (BF) asm volatile(IR_SP_FMA IR_SP_REST("v_cvt_pk_bf16_f32", "%[t0]", "%[t1]") : IR_SP_OUT : IR_SP_IN, [c2] "s"(c2), [nmc] "v"(nmc));
    else asm volatile(IR_SP_FMA IR_SP_REST("v_cvt_pk_f16_f32", "%[t0]", "%[t1]") : IR_SP_OUT : IR_SP_IN, [c2] "s"(c2), [nmc] "v"(nmc));
  }
#undef IR_SP_IN
#undef IR_SP_OUT
#undef IR_SP_FMA
#undef IR_SP_REST
}

struct Walker {   // one DMA stream (K or V) over the segment list, all wave-uniform except vo[]
  i32x4 rw;
  int step, sntile, seg, t0, issued;
  unsigned vo[2];
};

template <typename T, bool FOLD, bool PRESC>
__global__ void __launch_bounds__(256, 1) shared_attn_fwd_sp_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  using v2 = typename Pair2<T>::type;
  constexpr int NW = 4, NT = 256, QB = 256, CH = 2;
  constexpr int K_OFF = 0, V_OFF = KRING * TILE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(KRING + VRING) * TILE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lq = lane & 31;

  // ---- work decode (whole items, then K/V-range pieces of the remainder items) -------------------
  const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
  int item_local, piece = 0, npiece = 1;
  if (xslot < p.sk_full) {
    item_local = xslot;
  } else {
    npiece = p.sk_k;
    const int r = xslot - p.sk_full;
    item_local = p.sk_full + r / npiece;
    piece = r - (r / npiece) * npiece;
  }
  const int lin = xcd * p.sk_ix + item_local;
  if (item_local >= p.sk_ix || lin >= p.sk_items) return;
  const int tile_begin = (int)(((long)p.ntiles * piece) / npiece);
  const int tile_end = (int)(((long)p.ntiles * (piece + 1)) / npiece);
  const int NTILES = tile_end - tile_begin;
  const int bh = lin / p.nqb, qb = lin - bh * p.nqb;
  const int b = bh / p.H, h = bh - b * p.H;

  // ---- Q fragments of both row blocks (B operand of S^T = K Q^T) -----------------------------------
  const int qrowA = qb * QB + wid * 64 + lq, qrowB = qrowA + 32;
  v8 qA[4], qB[4];
  {
    const int ra = qrowA < p.Lq ? qrowA : p.Lq - 1, rb = qrowB < p.Lq ? qrowB : p.Lq - 1;
    const T* base = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qA[ks] = *(const v8*)(base + (int64_t)ra * p.q_sl + ks * 16);
      qB[ks] = *(const v8*)(base + (int64_t)rb * p.q_sl + ks * 16);
    }
  }

  // ---- LDS read offsets (the K-tile XOR swizzle and the V-tile half swap of shared_attn_fwd.hip) -----
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

  // ---- DMA streams (lane-linear LDS image, swizzle applied to the source slot) ---------------------
  const int pslot = tid & 7;
  int srow[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) srow[c] = (tid >> 3) + c * (NT / 8);
  const int nseg = p.include_self + p.N;
  auto setup = [&](Walker& W, int s, bool isK) {
    const T* base;
    int sl_b, slen;
    if (p.include_self && s == 0) {
      base = isK ? (const T*)p.k_self + (int64_t)b * p.ks_sb + (int64_t)h * p.ks_sh
                 : (const T*)p.v_self + (int64_t)b * p.vs_sb + (int64_t)h * p.vs_sh;
      sl_b = (int)(isK ? p.ks_sl : p.vs_sl) * 2; slen = p.Ls; W.sntile = p.tiles_self;
    } else {
      const int n = s - p.include_self;
      base = isK ? (const T*)p.k_ref + (int64_t)b * p.kr_sb + (int64_t)n * p.kr_sn + (int64_t)h * p.kr_sh
                 : (const T*)p.v_ref + (int64_t)b * p.vr_sb + (int64_t)n * p.vr_sn + (int64_t)h * p.vr_sh;
      sl_b = (int)(isK ? p.kr_sl : p.vr_sl) * 2; slen = p.Lr; W.sntile = p.tiles_ref;
    }
    W.rw = make_rsrc_words(base, (unsigned)((slen - 1) * sl_b + 128));   // rows past the segment end read as zeros
    W.step = KVB * sl_b;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int swz = isK ? ((srow[c] >> 1) & 7) : (((srow[c] >> 1) & 1) << 2);
      W.vo[c] = (unsigned)(srow[c] * sl_b + ((pslot ^ swz) * 16));
    }
  };
  auto issue = [&](Walker& W, bool isK, int slot) {
    if (W.issued >= NTILES) return;
    unsigned char* dst = smem + (isK ? K_OFF : V_OFF) + slot * TILE_BYTES;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      buffer_load_lds16_async(W.rw, dst + (c * NW + wid) * 1024, W.vo[c]);
      W.vo[c] += W.step;
    }
    ++W.issued;
    if (++W.t0 == W.sntile) {
      W.t0 = 0;
      if (++W.seg < nseg) setup(W, W.seg, isK);
    }
  };

  // first tile of this piece -> (segment, tile inside the segment)
  int seg_b = 0, t0_b = tile_begin;
  if (!(p.include_self && tile_begin < p.tiles_self)) {
    const int r = tile_begin - p.tiles_self;
    seg_b = p.include_self + r / p.tiles_ref;
    t0_b = r - (r / p.tiles_ref) * p.tiles_ref;
  }
  Walker WK, WV;
  WK.seg = WV.seg = seg_b; WK.t0 = WV.t0 = t0_b; WK.issued = WV.issued = 0;
  setup(WK, seg_b, true);
  setup(WV, seg_b, false);
#pragma unroll
  for (int c = 0; c < CH; ++c) { WK.vo[c] += (unsigned)(t0_b * WK.step); WV.vo[c] += (unsigned)(t0_b * WV.step); }

  // ---- per-row-block state ---------------------------------------------------------------------------
  f32x16 oA0, oA1, oB0, oB1;     // O^T accumulators (d = 32*db + crow(r,hi), column = query row)
#pragma unroll
  for (int r = 0; r < 16; ++r) { oA0[r] = 0.f; oA1[r] = 0.f; oB0[r] = 0.f; oB1[r] = 0.f; }
  float lA0 = 0.f, lA1 = 0.f, lB0 = 0.f, lB1 = 0.f;   // partial row sums of the current segment (whole run if !FOLD)
  float ldA = 0.f, ldB = 0.f;                          // FOLD: row sum of the finished segments
  // running (lazy) reference of the scores: raw-score units, or exponent units with PRESC (then also held,
  // negated, in a 16-register block that is the C operand of the first QK^T MFMA of every tile)
  float mA = PRESC ? 0.f : -INFINITY, mB = mA;
  f32x16 nmA, nmB;
#pragma unroll
  for (int r = 0; r < 16; ++r) { nmA[r] = 0.f; nmB[r] = 0.f; }
  const float c2 = PRESC ? 1.0f : p.scale_log2;
  const float lazy_thr = 6.0f / c2;   // keep the reference while no row's max grows by more than 2^6 (P <= 64)

  // compute-stream bookkeeping: tile i -> its segment, whether it closes the segment, its valid key count
  int cseg = seg_b, ct0 = t0_b;
  auto seg_tiles = [&](int s) { return (p.include_self && s == 0) ? p.tiles_self : p.tiles_ref; };
  auto seg_len = [&](int s) { return (p.include_self && s == 0) ? p.Ls : p.Lr; };
  auto tile_valid = [&](int s, int t) { const int v = seg_len(s) - t * KVB; return v < KVB ? v : KVB; };

  auto mask_tile = [&](f32x16& s0, f32x16& s1, int valid) {   // ragged last tile of a segment: keys >= valid -> -inf
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= valid) s0[r] = -INFINITY;
      if (key + 32 >= valid) s1[r] = -INFINITY;
    }
  };
  auto cross_max = [&](float mx) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), mx);
  };
  // lazy rescale of one row block; `mx` = row max of the NEXT tile to be exponentiated (raw, or relative to the
  // reference with PRESC); everything still at the old reference is scaled exactly once: O, the row sums, and with
  // PRESC the scores `s0/s1` that were produced against the old reference
  auto rescale = [&](float mx, float& m, f32x16& nm, f32x16& o0, f32x16& o1, float& l0, float& l1, float& ld,
                     f32x16& s0, f32x16& s1, bool force) {
    if (PRESC) {
      if (force || __any(mx > lazy_thr)) {
        const float d = force ? mx : max3(mx, 0.f, 0.f);
        const float alpha = fast_exp2(-d);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; s0[r] -= d; s1[r] -= d; }
        l0 *= alpha; l1 *= alpha;
        if (FOLD) ld *= alpha;
        m += d;
#pragma unroll
        for (int r = 0; r < 16; ++r) nm[r] = -m;
      }
    } else {
      if (__any(mx > m + lazy_thr)) {
        const float m_new = max3(m, mx, mx);
        const float alpha = fast_exp2((m - m_new) * c2);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        l0 *= alpha; l1 *= alpha;
        if (FOLD) ld *= alpha;
        m = m_new;
      }
    }
  };
  // FOLD: close segment `sc` of one row block: acc' <- acc' * (a_cur / a_next) + l_seg * (b_cur / a_next)
  auto fold_boundary = [&](int sc, bool has_next, f32x16& o0, f32x16& o1, float& l0, float& l1, float& ld) {
    float ls = l0 + l1;
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
    ls = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    ld += ls;
    l0 = 0.f; l1 = 0.f;
    const bool cur_ref = !(p.include_self && sc == 0);
    const int64_t ao_c = ((int64_t)(b * p.N + (cur_ref ? sc - p.include_self : 0)) * p.H + h) * 64 + 4 * hi;
    const int64_t ao_n = ((int64_t)(b * p.N + (has_next ? sc + 1 - p.include_self : 0)) * p.H + h) * 64 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 ac0 = {1.f, 1.f, 1.f, 1.f}, ac1 = ac0, an0 = ac0, an1 = ac0, bc0 = {0.f, 0.f, 0.f, 0.f}, bc1 = bc0;
      if (cur_ref) {
        ac0 = *(const f32x4*)(p.aa + ao_c + 8 * g4); ac1 = *(const f32x4*)(p.aa + ao_c + 32 + 8 * g4);
        bc0 = *(const f32x4*)(p.ab + ao_c + 8 * g4); bc1 = *(const f32x4*)(p.ab + ao_c + 32 + 8 * g4);
      }
      if (has_next) {
        an0 = *(const f32x4*)(p.aa + ao_n + 8 * g4); an1 = *(const f32x4*)(p.aa + ao_n + 32 + 8 * g4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        const float i0 = 1.0f / an0[i], i1 = 1.0f / an1[i];
        o0[r] = __builtin_fmaf(o0[r], ac0[i] * i0, ls * (bc0[i] * i0));
        o1[r] = __builtin_fmaf(o1[r], ac1[i] * i1, ls * (bc1[i] * i1));
      }
    }
  };

  // ---- fragment readers -----------------------------------------------------------------------------------
  auto k_frag = [&](const unsigned char* Kb, int half, int ks) -> v8 {
    return *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + half * 32 * 128 + kread[ks]);
  };
  auto v_frag = [&](const unsigned char* Vb, int db, int j) -> v8 {   // j = 2*kb + ks: keys 32kb + 16ks ..
    const int off = (32 * (j >> 1) + 16 * (j & 1)) * 128;
    return join_tr<v8>(lds_read_tr16(Vb + vread[db] + off), lds_read_tr16(Vb + vread[db] + off + 8 * 128));
  };

  // ---- prologue: tiles K(0..2), V(0); S(0) of both blocks; reference of block A ---------------------------
  issue(WK, true, 0);
  issue(WV, false, 0);
  issue(WK, true, 1);
  issue(WK, true, 2);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qA[ks]), "v"(qB[ks]));   // retire the Q loads here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(qA[ks]), "+a"(qB[ks]));   // Q lives in AGPRs from here on (MFMA operand only)

  f32x16 sa0x, sa1x, sb0x, sb1x;   // score blocks, buffer x
  f32x16 sa0y, sa1y, sb0y, sb1y;   // score blocks, buffer y
  v8 kf0[4], kf1[4];               // K fragments of the tile whose QK^T is next (keys 0..31 / 32..63)
  v8 vf0[4], vf1[4];               // V^T fragments of the tile whose P V of block B is pending
  unsigned wB[4][4];               // probabilities of block B's pending tile as packed 16-bit pairs: B operand of
                                   // O^T = V^T P^T, fragment j = 2*kb + ks (block A's live inside one step)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { vf0[j][e] = (T)0.f; vf1[j][e] = (T)0.f; }
#pragma unroll
    for (int e = 0; e < 4; ++e) wB[j][e] = 0u;
  }
  {
    const unsigned char* Kb = smem + K_OFF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf0[ks] = k_frag(Kb, 0, ks); kf1[ks] = k_frag(Kb, 1, ks); }
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa0x[r] = 0.f; sa1x[r] = 0.f; sb0x[r] = 0.f; sb1x[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      sa0x = Tr::mfma(kf0[ks], qA[ks], sa0x);
      sa1x = Tr::mfma(kf1[ks], qA[ks], sa1x);
      sb0x = Tr::mfma(kf0[ks], qB[ks], sb0x);
      sb1x = Tr::mfma(kf1[ks], qB[ks], sb1x);
    }
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(sa0x), "+v"(sa1x), "+v"(sb0x), "+v"(sb1x));  // MFMA -> asm v_max3 pad
    const int valid0 = tile_valid(cseg, ct0);
    if (valid0 < KVB) { mask_tile(sa0x, sa1x, valid0); mask_tile(sb0x, sb1x, valid0); }
    float mx = max3(sa0x[0], sa0x[1], sa1x[0]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = max3(mx, sa0x[r], sa0x[r + 1]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3(mx, sa1x[r], sa1x[r + 1]);
    mx = cross_max(max3(mx, sa1x[15], sa1x[15]));
    rescale(mx, mA, nmA, oA0, oA1, lA0, lA1, ldA, sa0x, sa1x, true);
    if (PRESC) {   // block B: same forced start, its max is taken here too (X(0) would compare against m = 0)
      float my = max3(sb0x[0], sb0x[1], sb1x[0]);
#pragma unroll
      for (int r = 2; r < 16; r += 2) my = max3(my, sb0x[r], sb0x[r + 1]);
#pragma unroll
      for (int r = 1; r < 15; r += 2) my = max3(my, sb1x[r], sb1x[r + 1]);
      my = cross_max(max3(my, sb1x[15], sb1x[15]));
      rescale(my, mB, nmB, oB0, oB1, lB0, lB1, ldB, sb0x, sb1x, true);
    }
    // K fragments of tile 1 (stale but harmless LDS contents if the piece has a single tile)
    const unsigned char* K1 = smem + K_OFF + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) { kf0[ks] = k_frag(K1, 0, ks); kf1[ks] = k_frag(K1, 1, ks); }
  }

  // ---- one tile step ---------------------------------------------------------------------------------------
  auto sreg = [](f32x16& s0, f32x16& s1, int R) -> float { return R < 16 ? s0[R] : s1[R - 16]; };

  int kslot_rd = 2;   // K ring slot of tile i+2 (its fragments are read in Y(i)); tile j lives in slot j % 3
  int kslot_wr = 0;   // slot of tile i+3 (= the slot tile i left), written at the start of iteration i
  int prev_seg = seg_b;
  bool prev_last = false;
  auto step = [&](int i, f32x16& sa0c, f32x16& sa1c, f32x16& sa0n, f32x16& sa1n,
                  f32x16& sb0c, f32x16& sb1c, f32x16& sb0n, f32x16& sb1n) {
    if (!(SP_ABL & 2)) {
      issue(WK, true, kslot_wr);          // K(i+3)
      issue(WV, false, (i + 1) & 1);      // V(i+1)
    }
    const unsigned char* Vb = smem + V_OFF + (i & 1) * TILE_BYTES;
    const unsigned char* Kn = smem + K_OFF + kslot_rd * TILE_BYTES;
    const bool more = i + 1 < NTILES;
    // bookkeeping of tiles i (current) and i+1 (next)
    const bool cur_last = (ct0 + 1 == seg_tiles(cseg));
    const int nseg_i = cur_last ? cseg + 1 : cseg, nt0_i = cur_last ? 0 : ct0 + 1;
    const int valid_next = more ? tile_valid(nseg_i, nt0_i) : KVB;

    // ===== region X(i): MFMA QK_A(i+1), PV_B(i-1) | VALU exp of S_A(i), max of S_B(i) | LDS V(i) fragments =====
    const float nmcA = -mA * c2;
    v8 vn0[4], vn1[4];
    float mxB = -INFINITY;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // K(i+1) fragments (read during Y(i-1)) feed asm MFMAs below
    unsigned wA[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x16& se = (j < 2) ? sa0c : sa1c;
      const int rb = (8 * j) & 15;
      const v8 pB = __builtin_bit_cast(v8, u32x4{wB[j][0], wB[j][1], wB[j][2], wB[j][3]});
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u == 0) { if (j == 0) sp_mfma<T, false, PRESC ? 2 : 1, true>(sa0n, kf0[j], qA[j], nmA); else sp_mfma<T, false, 0, true>(sa0n, kf0[j], qA[j], nmA); }
        if (u == 1) { if (j == 0) sp_mfma<T, false, PRESC ? 2 : 1, true>(sa1n, kf1[j], qA[j], nmA); else sp_mfma<T, false, 0, true>(sa1n, kf1[j], qA[j], nmA); }
#if SP_PV_MODE == 1
        if (u == 2) oB0 = Tr::mfma(vf0[j], pB, oB0);
        if (u == 3) oB1 = Tr::mfma(vf1[j], pB, oB1);
#elif SP_PV_MODE == 2
        if (u == 2) sp_mfma<T, false, 0, false>(oB0, vf0[j], pB, nmA);
        if (u == 3) sp_mfma<T, false, 0, false>(oB1, vf1[j], pB, nmA);
#else
        if (u == 2) sp_mfma<T, true, 0, false>(oB0, vf0[j], pB, nmA);
        if (u == 3) sp_mfma<T, true, 0, false>(oB1, vf1[j], pB, nmA);
#endif
        const int R = 2 * (4 * j + u);
        sp_vec<T, PRESC>(se[rb + 2 * u], se[rb + 2 * u + 1], c2, nmcA, lA0, lA1, wA[j][u], mxB, sreg(sb0c, sb1c, R), sreg(sb0c, sb1c, R + 1), pB);
        if (SP_ABL & 4) { vn0[j] = vf0[j]; vn1[j] = vf1[j]; }
        else {
          if (u == 2) vn0[j] = v_frag(Vb, 0, j);
          if (u == 3) vn1[j] = v_frag(Vb, 1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- boundary 1: P V of block B for tile i-1 is complete --------------------------------------------------
    if (FOLD && prev_last) fold_boundary(prev_seg, true, oB0, oB1, lB0, lB1, ldB);
    if (!(SP_ABL & 8)) {
      mxB = cross_max(mxB);
      rescale(mxB, mB, nmB, oB0, oB1, lB0, lB1, ldB, sb0c, sb1c, false);
    }
    if (!(SP_ABL & 64)) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(sa0n), "+v"(sa1n));   // QK_A MFMA results -> VALU (mask below, max in Y)
    if (valid_next < KVB) mask_tile(sa0n, sa1n, valid_next);

    // ===== region Y(i): MFMA QK_B(i+1), PV_A(i) | VALU exp of S_B(i), max of S_A(i+1) | LDS K(i+2) fragments =====
    const float nmcB = -mB * c2;
    v8 kn0[4], kn1[4];
    float mxA = -INFINITY;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // V(i) fragments (read during X) feed asm MFMAs below
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x16& se = (j < 2) ? sb0c : sb1c;
      const int rb = (8 * j) & 15;
      const v8 pA = __builtin_bit_cast(v8, u32x4{wA[j][0], wA[j][1], wA[j][2], wA[j][3]});
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u == 0) { if (j == 0) sp_mfma<T, false, PRESC ? 2 : 1, true>(sb0n, kf0[j], qB[j], nmB); else sp_mfma<T, false, 0, true>(sb0n, kf0[j], qB[j], nmB); }
        if (u == 1) { if (j == 0) sp_mfma<T, false, PRESC ? 2 : 1, true>(sb1n, kf1[j], qB[j], nmB); else sp_mfma<T, false, 0, true>(sb1n, kf1[j], qB[j], nmB); }
#if SP_PV_MODE == 1
        if (u == 2) oA0 = Tr::mfma(vn0[j], pA, oA0);
        if (u == 3) oA1 = Tr::mfma(vn1[j], pA, oA1);
#elif SP_PV_MODE == 2
        if (u == 2) sp_mfma<T, false, 0, false>(oA0, vn0[j], pA, nmB);
        if (u == 3) sp_mfma<T, false, 0, false>(oA1, vn1[j], pA, nmB);
#else
        if (u == 2) sp_mfma<T, true, 0, false>(oA0, vn0[j], pA, nmB);
        if (u == 3) sp_mfma<T, true, 0, false>(oA1, vn1[j], pA, nmB);
#endif
        const int R = 2 * (4 * j + u);
        sp_vec<T, PRESC>(se[rb + 2 * u], se[rb + 2 * u + 1], c2, nmcB, lB0, lB1, wB[j][u], mxA, sreg(sa0n, sa1n, R), sreg(sa0n, sa1n, R + 1), pA);
        if (SP_ABL & 4) { kn0[j] = kf0[j]; kn1[j] = kf1[j]; }
        else {
          if (u == 2) kn0[j] = k_frag(Kn, 0, j);
          if (u == 3) kn1[j] = k_frag(Kn, 1, j);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!(SP_ABL & 64)) asm volatile("s_nop 7\n\ts_nop 4" : "+v"(sb0n), "+v"(sb1n));   // QK_B MFMA results -> VALU
    // ---- boundary 2: P V of block A for tile i is complete ----------------------------------------------------
    if (FOLD && (cur_last || !more)) fold_boundary(cseg, more, oA0, oA1, lA0, lA1, ldA);
    if (more && !(SP_ABL & 8)) {
      mxA = cross_max(mxA);
      rescale(mxA, mA, nmA, oA0, oA1, lA0, lA1, ldA, sa0n, sa1n, false);
      if (valid_next < KVB) mask_tile(sb0n, sb1n, valid_next);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { kf0[j] = kn0[j]; kf1[j] = kn1[j]; vf0[j] = vn0[j]; vf1[j] = vn1[j]; }
    prev_last = cur_last; prev_seg = cseg;
    cseg = nseg_i; ct0 = nt0_i;
    kslot_rd = kslot_rd == 2 ? 0 : kslot_rd + 1;
    kslot_wr = kslot_wr == 2 ? 0 : kslot_wr + 1;
    if (!(SP_ABL & 1)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this iteration's transfers were issued a tile ago
      __syncthreads();
    }
  };

  int i = 0;
  for (; i + 1 < NTILES; i += 2) {
    step(i, sa0x, sa1x, sa0y, sa1y, sb0x, sb1x, sb0y, sb1y);
    step(i + 1, sa0y, sa1y, sa0x, sa1x, sb0y, sb1y, sb0x, sb1x);
  }
  if (i < NTILES) step(i, sa0x, sa1x, sa0y, sa1y, sb0x, sb1x, sb0y, sb1y);

  // ---- drain: P V of block B for the last tile ---------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const v8 pB = __builtin_bit_cast(v8, u32x4{wB[j][0], wB[j][1], wB[j][2], wB[j][3]});
    sp_mfma<T, true, 0, false>(oB0, vf0[j], pB, nmB);
    sp_mfma<T, true, 0, false>(oB1, vf1[j], pB, nmB);
  }
  asm volatile("s_nop 7\n\ts_nop 4" : "+a"(oB0), "+a"(oB1));
  if (FOLD) fold_boundary(prev_seg, false, oB0, oB1, lB0, lB1, ldB);

  // ---- epilogue (per row block) ---------------------------------------------------------------------------------
  auto finish = [&](f32x16& o0, f32x16& o1, float l0, float l1, float ld, float m, int qrow, int rowoff) {
    float l_fin;
    if (FOLD) {
      l_fin = ld;
    } else {
      float ls = l0 + l1;
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
      l_fin = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float m_raw = PRESC ? m / p.scale_log2 : m;   // the combine kernel and the LSE work in raw-score units
    if (npiece > 1) {
      const int64_t prow = ((int64_t)((xcd * (p.sk_ix - p.sk_full) + (item_local - p.sk_full)) * npiece + piece)) * QB + wid * 64 + rowoff + lq;
      float* wo = p.ws_o + prow * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 x0, x1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x0[e] = o0[4 * g4 + e]; x1[e] = o1[4 * g4 + e]; }
        *(f32x4*)(wo + 8 * g4 + 4 * hi) = x0;
        *(f32x4*)(wo + 32 + 8 * g4 + 4 * hi) = x1;
      }
      if (hi == 0) {
        p.ws_ml[prow * 2] = m_raw;
        p.ws_ml[prow * 2 + 1] = l_fin;
      }
      return;
    }
    const float inv = 1.0f / l_fin;
    if (qrow < p.Lq) {
      const int64_t off = (int64_t)b * p.o_sb + (int64_t)qrow * p.o_sl + (int64_t)h * p.o_sh;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 x0, x1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { x0[e] = o0[4 * g4 + e] * inv; x1[e] = o1[4 * g4 + e] * inv; }
        if (p.out_f32) {
          float* op = (float*)p.out + off;
          *(f32x4*)(op + 8 * g4 + 4 * hi) = x0;
          *(f32x4*)(op + 32 + 8 * g4 + 4 * hi) = x1;
        } else {
          T* op = (T*)p.out + off;
          *(v4*)(op + 8 * g4 + 4 * hi) = __builtin_convertvector(x0, v4);
          *(v4*)(op + 32 + 8 * g4 + 4 * hi) = __builtin_convertvector(x1, v4);
        }
      }
      if (p.lse != nullptr && hi == 0)
        p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_raw * p.scale + __logf(l_fin);
    }
  };
  finish(oA0, oA1, lA0, lA1, ldA, mA, qrowA, 0);
  finish(oB0, oB1, lB0, lB1, ldB, mB, qrowB, 32);
}

template <typename T, bool FOLD, bool PRESC>
hipError_t launch(const AttnKParams& p0, hipStream_t s) {
  AttnKParams p = p0;
  constexpr int QB = 256;
  p.nqb = (p.Lq + QB - 1) / QB;
  p.sk_items = p.B * p.H * p.nqb;
  p.sk_ix = (p.sk_items + 7) / 8;
  const int slots_x = 32;   // one 4-wave workgroup per CU, 32 CUs per XCD
  int full = (p.sk_ix / slots_x) * slots_x;
  int rem = p.sk_ix - full;
  int k = 1;
  if (p.ws != nullptr && rem > 0) {
    const size_t piece_bytes = (size_t)QB * 66 * sizeof(float);
    k = ir_pick_split(rem, slots_x, p.ntiles / 8 /* pieces of at least 8 tiles */, (long)(p.ws_bytes / piece_bytes / 8));
  }
  if (k <= 1) { full = p.sk_ix; rem = 0; k = 1; }
  p.sk_full = full;
  p.sk_k = k;
  p.ws_o = p.ws;
  p.ws_ml = p.ws + (size_t)8 * rem * k * QB * 64;
  const int grid = 8 * (full + rem * k);
  hipLaunchKernelGGL((shared_attn_fwd_sp_kernel<T, FOLD, PRESC>), dim3(grid), dim3(256), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || k <= 1) return e;
  return ir_launch_shared_attn_combine(p, std::is_same<T, __bf16>::value ? 1 : 0, QB, rem, s);
}

template <typename T>
hipError_t launch_t(const AttnKParams& p, hipStream_t s) {
  const bool fold = p.aa != nullptr;
  // the PRESC template path of this kernel is not instantiated: it failed the pre-scaled-Q contract test on multi-tile
  // walks (and ran slower: two more 16-register blocks push the allocator into AGPR copies); pre-scaled Q is served by the
  // 64-row and 32-row kernels (shared_attn_fwd.hip)
  if (p.q_prescaled) return hipErrorInvalidValue;
  return fold ? launch<T, true, false>(p, s) : launch<T, false, false>(p, s);
}

}  // namespace

hipError_t ir_launch_shared_attn_fwd_sp(const AttnKParams& p, int dtype, hipStream_t s) {
  return dtype == 1 ? launch_t<__bf16>(p, s) : launch_t<_Float16>(p, s);
}

#endif  // IR_ABLATIONS
