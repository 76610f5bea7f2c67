// shared_attn_fwd_tp.hip - 32 query rows per wave, TWO waves per SIMD, three-stage software pipeline with a
// spelled-out MFMA / VALU interleave (gfx950).  Same math, layouts and C-ABI contract as the other kernels
// (shared_attn_fwd.hip has the derivation).
//
// What round 2 measured (profiles/r2_ubench_gapfill.txt, r2_sp_ablation.txt, r2_kernel_experiments.txt):
//   * the matrix pipe and the VALU only overlap when EACH wave's own instruction stream interleaves them (a wave
//     issuing MFMAs back to back keeps the issue port; phase offsets between waves change nothing);
//   * one wave hides ~4.5 single-issue VALU instructions per MFMA gap, two waves per SIMD that both interleave run
//     at ~51 cycles per MFMA with 8 fillers per gap - the 64-row kernel needs ~75;
//   * with ONE wave per SIMD (shared_attn_fwd_sp.hip) every LDS wait, DMA issue and barrier is exposed.
// So: two waves per SIMD (<= 256 registers: one 32-row block per wave), and a pipeline in which NO vector
// instruction of an iteration depends on an MFMA of the same iteration:
//       iteration i :   MFMA   QK^T(i+1)  (clusters 0-7)     PV(i-1)  (clusters 8-15)
//                       VALU   exp / row sum / convert of S(i)  (one score pair per cluster)
//                              row max of S(i+1)                (clusters 10-15, after its last MFMA has retired)
// (the row max occupies clusters 10-13: two chains, four registers each per cluster)
// The rescale decision for tile i+1 falls at the end of iteration i, when P(i) has been exponentiated against the old
// reference but not yet multiplied into O (P V lags by one iteration): the row sums are rescaled at once (they already
// hold P(i)), O one iteration later, after P(i) V(i) has been added (`alpha_pend`) - everything at the old reference is
// scaled exactly once.  The AdaIN segment boundary is delayed the same way (`fold_pend`).
// Each cluster = two adjacent volatile asm statements (see shared_attn_fwd_sp.hip for why not builtins).
// 8 waves = 256 query rows per workgroup, one workgroup per CU; K and V rings of 2 tiles, LDS-DMA one iteration ahead.
#ifdef IR_ABLATIONS   // documented experiment (variant 17, DESIGN.md 4.1b'): development builds only

#include <type_traits>

#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int KVB = IR_KV_TILE;
constexpr int TILE_BYTES = KVB * 64 * 2;  // 8 KiB
// TP_ABL: timing ablations (development only, WRONG results): 1 no barrier, 2 no DMA, 4 no LDS fragment reads,
// 8 no boundary work, 16 no vector work, 32 no MFMAs
#ifndef TP_ABL
#define TP_ABL 0
#endif

// CMODE: 0 accumulate, 1 C = 0, 2 C = `cblk` (PRESC: minus the running reference)
template <typename T, int CMODE, typename V8>
__device__ __forceinline__ void tp_mfma(f32x16& acc, const V8& a, const V8& b, const f32x16& cblk) {
  constexpr bool BF = std::is_same<T, __bf16>::value;
  if (TP_ABL & 32) { asm volatile("" : "+v"(acc) : "v"(a), "v"(b)); return; }
  if (CMODE == 0) {
    if (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  } else if (CMODE == 1) {
    if (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
  } else {
    if (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(cblk));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(cblk));
  }
}

// vector work of one cluster: two scores of the tile being exponentiated and NMAX (0..2) steps of each of the two
// row-max chains of the tile that follows.  `ha`, `hb`: the operands of the MFMA issued just before - listed as unused
// inputs so that the allocator does not hand their registers to this block's temporaries while the matrix pipe may
// still be reading them.
template <typename T, bool PRESC, int NMAX, typename V8>
__device__ __forceinline__ void tp_vec(float s0, float s1, float c2, float nmc, float& l0, float& l1, unsigned& pk,
                                       float& mxa, float& mxb, float a0, float a1, float b0, float b1, float a2, float a3,
                                       float b2, float b3, const V8& ha, const V8& hb) {
  constexpr bool BF = std::is_same<T, __bf16>::value;
  if (TP_ABL & 16) { asm volatile("" : "+v"(l0), "+v"(l1), "=v"(pk), "+v"(mxa), "+v"(mxb) : "v"(s0), "v"(s1), "v"(ha), "v"(hb)); return; }
  float t0, t1;
#define TP_FMA "v_fma_f32 %[t0], %[s0], %[c2], %[nmc]\n\tv_fma_f32 %[t1], %[s1], %[c2], %[nmc]\n\t"
#define TP_EXP(X0, X1) "v_exp_f32 %[t0], " X0 "\n\tv_exp_f32 %[t1], " X1 "\n\t"
#define TP_MAX1 "v_max3_f32 %[mxa], %[mxa], %[a0], %[a1]\n\tv_max3_f32 %[mxb], %[mxb], %[b0], %[b1]\n\t"
#define TP_MAX2 "v_max3_f32 %[mxa], %[mxa], %[a2], %[a3]\n\tv_max3_f32 %[mxb], %[mxb], %[b2], %[b3]\n\t"
#define TP_SPACER "s_nop 0\n\t"   /* exp -> add two instructions apart when no max sits between them */
#define TP_TAIL(CVT) "v_add_f32 %[l0], %[l0], %[t0]\n\tv_add_f32 %[l1], %[l1], %[t1]\n\t" CVT " %[pk], %[t0], %[t1]"
#define TP_OUT [t0] "=&v"(t0), [t1] "=&v"(t1), [l0] "+v"(l0), [l1] "+v"(l1), [pk] "=v"(pk), [mxa] "+v"(mxa), [mxb] "+v"(mxb)
#define TP_IN [s0] "v"(s0), [s1] "v"(s1), [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [a2] "v"(a2), [a3] "v"(a3), \
              [b2] "v"(b2), [b3] "v"(b3), [ha] "v"(ha), [hb] "v"(hb), [c2] "s"(c2), [nmc] "v"(nmc)
#define TP_BODY(CVT)                                                                                              \
  if (PRESC) {                                                                                                    \
    if (NMAX == 0) asm volatile(TP_EXP("%[s0]", "%[s1]") TP_SPACER TP_TAIL(CVT) : TP_OUT : TP_IN);                \
    else if (NMAX == 1) asm volatile(TP_EXP("%[s0]", "%[s1]") TP_MAX1 TP_TAIL(CVT) : TP_OUT : TP_IN);             \
    else asm volatile(TP_EXP("%[s0]", "%[s1]") TP_MAX1 TP_MAX2 TP_TAIL(CVT) : TP_OUT : TP_IN);                    \
  } else {                                                                                                        \
    if (NMAX == 0) asm volatile(TP_FMA TP_EXP("%[t0]", "%[t1]") TP_SPACER TP_TAIL(CVT) : TP_OUT : TP_IN);         \
    else if (NMAX == 1) asm volatile(TP_FMA TP_EXP("%[t0]", "%[t1]") TP_MAX1 TP_TAIL(CVT) : TP_OUT : TP_IN);      \
    else asm volatile(TP_FMA TP_EXP("%[t0]", "%[t1]") TP_MAX1 TP_MAX2 TP_TAIL(CVT) : TP_OUT : TP_IN);             \
  }
  if (BF) { TP_BODY("v_cvt_pk_bf16_f32") } else { TP_BODY("v_cvt_pk_f16_f32") }
#undef TP_BODY
#undef TP_IN
#undef TP_OUT
#undef TP_TAIL
#undef TP_SPACER
#undef TP_MAX2
#undef TP_MAX1
#undef TP_EXP
#undef TP_FMA
}

struct TpWalker {   // one DMA stream (K or V) over the segment list; wave-uniform except vo
  i32x4 rw;
  int step, sntile, seg, t0, issued;
  unsigned vo;
};

template <typename T, bool FOLD, bool PRESC>
__global__ void __launch_bounds__(512, 1) shared_attn_fwd_tp_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NW = 8, QB = 256;
  constexpr int K_OFF = 0, V_OFF = 2 * TILE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE_BYTES];   // K ring of 2, V ring of 2

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

  // ---- Q fragments (B operand of S^T = K Q^T) ------------------------------------------------------
  const int qrow = qb * QB + wid * 32 + lq;
  v8 qf[4];
  {
    const int rc = qrow < p.Lq ? qrow : p.Lq - 1;
    const T* base = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh + (int64_t)rc * p.q_sl + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8*)(base + ks * 16);
  }

  // ---- LDS read offsets (K-tile XOR swizzle / V-tile half swap of shared_attn_fwd.hip) ---------------
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

  // ---- DMA streams: 512 threads x 16 B = one 8-KiB tile per instruction round -------------------------
  const int pslot = tid & 7, srow = tid >> 3;
  const int nseg = p.include_self + p.N;
  auto setup = [&](TpWalker& W, int s, bool isK) {
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
    const int swz = isK ? ((srow >> 1) & 7) : (((srow >> 1) & 1) << 2);
    W.vo = (unsigned)(srow * sl_b + ((pslot ^ swz) * 16));
  };
  auto issue = [&](TpWalker& W, bool isK, int slot) {
    if (W.issued >= NTILES) return;
    buffer_load_lds16_async(W.rw, smem + (isK ? K_OFF : V_OFF) + slot * TILE_BYTES + wid * 1024, W.vo);
    W.vo += W.step;
    ++W.issued;
    if (++W.t0 == W.sntile) {
      W.t0 = 0;
      if (++W.seg < nseg) setup(W, W.seg, isK);
    }
  };

  int seg_b = 0, t0_b = tile_begin;
  if (!(p.include_self && tile_begin < p.tiles_self)) {
    const int r = tile_begin - p.tiles_self;
    seg_b = p.include_self + r / p.tiles_ref;
    t0_b = r - (r / p.tiles_ref) * p.tiles_ref;
  }
  TpWalker WK, WV;
  WK.seg = WV.seg = seg_b; WK.t0 = WV.t0 = t0_b; WK.issued = WV.issued = 0;
  setup(WK, seg_b, true);
  setup(WV, seg_b, false);
  WK.vo += (unsigned)(t0_b * WK.step);
  WV.vo += (unsigned)(t0_b * WV.step);

  // ---- state ---------------------------------------------------------------------------------------------
  f32x16 o0, o1;                 // O^T accumulators (d = 32*db + crow(r,hi), column = query row)
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float l0 = 0.f, l1 = 0.f;      // partial row sums of the current segment (whole run if !FOLD)
  float l_done = 0.f;            // FOLD: row sum of the segments already folded
  float m_run = PRESC ? 0.f : -INFINITY;   // running (lazy) reference: raw-score units, exponent units with PRESC
  f32x16 nm;                     // PRESC: minus the reference in all 16 registers (C operand of a tile's first QK^T MFMAs)
#pragma unroll
  for (int r = 0; r < 16; ++r) nm[r] = 0.f;
  const float c2 = PRESC ? 1.0f : p.scale_log2;
  const float lazy_thr = 6.0f / c2;
  float alpha_pend = 1.0f;       // factor O still owes to the last reference change (applied after the pending P V)
  bool fold_pend = false;        // FOLD: a segment closed in the previous iteration waits for its last P V
  int fold_seg = 0;
  bool fold_next = false;
  float ls_pend = 0.f;

  auto seg_tiles = [&](int s) { return (p.include_self && s == 0) ? p.tiles_self : p.tiles_ref; };
  auto seg_len = [&](int s) { return (p.include_self && s == 0) ? p.Ls : p.Lr; };
  auto tile_valid = [&](int s, int t) { const int v = seg_len(s) - t * KVB; return v < KVB ? v : KVB; };
  auto mask_tile = [&](f32x16& s0, f32x16& s1, int valid) {
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
  auto row_sum = [&]() {
    float ls = l0 + l1;
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  };
  // decision for the tile whose scores are in (s0, s1): moves the reference if some row's max grew too much; the row sums
  // (which already contain every exponentiated tile) are rescaled here, O later (alpha_pend)
  auto decide = [&](float mx, f32x16& s0, f32x16& s1, bool force) {
    if (PRESC) {
      if (force || __any(mx > lazy_thr)) {
        const float d = force ? mx : max3(mx, 0.f, 0.f);
        const float alpha = force ? 1.f : fast_exp2(-d);   // first tile: nothing accumulated yet, and 2^-d may be inf
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] -= d; s1[r] -= d; }
        l0 *= alpha; l1 *= alpha;
        if (FOLD) { l_done *= alpha; ls_pend *= alpha; }
        alpha_pend *= alpha;
        m_run += d;
#pragma unroll
        for (int r = 0; r < 16; ++r) nm[r] = -m_run;
      }
    } else {
      if (__any(mx > m_run + lazy_thr)) {
        const float m_new = max3(m_run, mx, mx);
        const float alpha = fast_exp2((m_run - m_new) * c2);
        l0 *= alpha; l1 *= alpha;
        if (FOLD) { l_done *= alpha; ls_pend *= alpha; }
        alpha_pend *= alpha;
        m_run = m_new;
      }
    }
  };
  auto apply_alpha = [&]() {   // O catches up with the reference (everything in it was accumulated against the old one)
    if (__any(alpha_pend != 1.0f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha_pend; o1[r] *= alpha_pend; }
    }
    alpha_pend = 1.0f;
  };
  // FOLD: close segment `sc` (ratio frame, see shared_attn_fwd_w64.hip): acc' <- acc' (a_cur / a_next) + l_seg (b_cur / a_next)
  auto fold_boundary = [&](int sc, bool has_next, float ls) {
    l_done += ls;
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
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const float i0 = 1.0f / an0[e], i1 = 1.0f / an1[e];
        o0[r] = __builtin_fmaf(o0[r], ac0[e] * i0, ls * (bc0[e] * i0));
        o1[r] = __builtin_fmaf(o1[r], ac1[e] * i1, ls * (bc1[e] * i1));
      }
    }
  };

  auto k_frag = [&](const unsigned char* Kb, int half, int ks) -> v8 {
    return *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + half * 32 * 128 + kread[ks]);
  };
  auto v_frag = [&](const unsigned char* Vb, int db, int j) -> v8 {   // j = 2*kb + ks: keys 32kb + 16ks ..
    const int off = (32 * (j >> 1) + 16 * (j & 1)) * 128;
    return join_tr<v8>(lds_read_tr16(Vb + vread[db] + off), lds_read_tr16(Vb + vread[db] + off + 8 * 128));
  };
  auto sreg = [](f32x16& s0, f32x16& s1, int R) -> float { return R < 16 ? s0[R] : s1[R - 16]; };

  // ---- prologue: K(0), K(1); S(0); its reference ------------------------------------------------------------
  issue(WK, true, 0);
  issue(WK, true, 1);
  // iteration 0 multiplies the (all-zero) probabilities of "tile -1" with whatever V slot 1 holds: make that zeros
  *(u32x4*)(smem + V_OFF + TILE_BYTES + tid * 16) = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));   // retire the Q loads here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 s0x, s1x, s0y, s1y;         // score blocks, buffers x / y (keys 0..31 / 32..63 of a tile)
  unsigned wx[4][4], wy[4][4];       // probabilities as packed 16-bit pairs, fragment j = 2*kb + ks
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { wx[j][e] = 0u; wy[j][e] = 0u; }
  int cseg = seg_b, ct0 = t0_b;      // segment / tile-in-segment of the tile that iteration i exponentiates
  {
    const unsigned char* Kb = smem + K_OFF;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0x[r] = 0.f; s1x[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0x = Tr::mfma(k_frag(Kb, 0, ks), qf[ks], s0x);
      s1x = Tr::mfma(k_frag(Kb, 1, ks), qf[ks], s1x);
    }
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(s0x), "+v"(s1x));   // MFMA -> asm v_max3 pad
    const int valid0 = tile_valid(cseg, ct0);
    if (valid0 < KVB) mask_tile(s0x, s1x, valid0);
    float mx = max3(s0x[0], s0x[1], s1x[0]);
#pragma unroll
    for (int r = 2; r < 16; r += 2) mx = max3(mx, s0x[r], s0x[r + 1]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) mx = max3(mx, s1x[r], s1x[r + 1]);
    mx = cross_max(max3(mx, s1x[15], s1x[15]));
    decide(mx, s0x, s1x, true);
    alpha_pend = 1.0f;   // nothing is in O yet
  }
  __syncthreads();       // every wave has read K(0): its slot may be refilled

  // ---- one tile step ------------------------------------------------------------------------------------------
  // sc/sn: scores of tile i (exponentiated here) / tile i+1 (produced here); wp/wc: probabilities of tile i-1 (consumed
  // by P V here) / tile i (produced here)
  auto step = [&](int i, f32x16& s0c, f32x16& s1c, f32x16& s0n, f32x16& s1n, unsigned (&wp)[4][4], unsigned (&wc)[4][4]) {
    if (!(TP_ABL & 2)) {
      issue(WK, true, i & 1);        // K(i+2) -> the slot K(i) left
      issue(WV, false, i & 1);       // V(i)   -> the slot V(i-2) left
    }
    const unsigned char* Kb = smem + K_OFF + ((i + 1) & 1) * TILE_BYTES;   // K(i+1)
    const unsigned char* Vb = smem + V_OFF + ((i + 1) & 1) * TILE_BYTES;   // V(i-1)
    const bool more = i + 1 < NTILES;
    const bool cur_last = (ct0 + 1 == seg_tiles(cseg));
    const int nseg_i = cur_last ? cseg + 1 : cseg, nt0_i = cur_last ? 0 : ct0 + 1;
    const int valid_next = more ? tile_valid(nseg_i, nt0_i) : KVB;
    const float nmc = -m_run * c2;
    float mxa = -INFINITY, mxb = -INFINITY;
    v8 kf0 = k_frag(Kb, 0, 0), kf1 = k_frag(Kb, 1, 0);
    // clusters 0-7: QK^T(i+1), K fragments one contraction step ahead
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      v8 kn0 = kf0, kn1 = kf1;
      if (ks < 3 && !(TP_ABL & 4)) { kn0 = k_frag(Kb, 0, ks + 1); kn1 = k_frag(Kb, 1, ks + 1); }
      {
        const int c = 2 * ks, R = 2 * c;
        if (ks == 0) tp_mfma<T, PRESC ? 2 : 1>(s0n, kf0, qf[ks], nm); else tp_mfma<T, 0>(s0n, kf0, qf[ks], nm);
        tp_vec<T, PRESC, 0>(sreg(s0c, s1c, R), sreg(s0c, s1c, R + 1), c2, nmc, l0, l1, wc[c >> 2][c & 3], mxa, mxb,
                            0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, kf0, qf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const int c = 2 * ks + 1, R = 2 * c;
        if (ks == 0) tp_mfma<T, PRESC ? 2 : 1>(s1n, kf1, qf[ks], nm); else tp_mfma<T, 0>(s1n, kf1, qf[ks], nm);
        tp_vec<T, PRESC, 0>(sreg(s0c, s1c, R), sreg(s0c, s1c, R + 1), c2, nmc, l0, l1, wc[c >> 2][c & 3], mxa, mxb,
                            0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, kf1, qf[ks]);
        __builtin_amdgcn_sched_barrier(0);
      }
      kf0 = kn0; kf1 = kn1;
    }
    // clusters 8-15: P V of tile i-1, V^T fragments one step ahead; the row max of tile i+1 rides in clusters 10-15
    v8 vf0 = v_frag(Vb, 0, 0), vf1 = v_frag(Vb, 1, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v8 vn0 = vf0, vn1 = vf1;
      if (j < 3 && !(TP_ABL & 4)) { vn0 = v_frag(Vb, 0, j + 1); vn1 = v_frag(Vb, 1, j + 1); }
      const v8 pp = __builtin_bit_cast(v8, u32x4{wp[j][0], wp[j][1], wp[j][2], wp[j][3]});
      if (j == 1) {
        // the last QK^T MFMA of tile i+1 was issued three clusters ago: its scores may be read (and, rarely, masked)
        asm volatile("" : "+v"(s0n), "+v"(s1n));
        if (valid_next < KVB) {
          asm volatile("s_nop 7\n\ts_nop 4" : "+v"(s0n), "+v"(s1n));
          mask_tile(s0n, s1n, valid_next);
        }
      }
      {
        const int c = 8 + 2 * j, R = 2 * c;
        tp_mfma<T, 0>(o0, vf0, pp, nm);
        if (j == 0 || j == 3) {
          tp_vec<T, PRESC, 0>(sreg(s0c, s1c, R), sreg(s0c, s1c, R + 1), c2, nmc, l0, l1, wc[c >> 2][c & 3], mxa, mxb,
                              0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, vf0, pp);
        } else {   // j = 1, 2: the two max chains (keys 0..31 / 32..63) take four registers each per cluster
          const int q4 = 8 * (j - 1);
          tp_vec<T, PRESC, 2>(sreg(s0c, s1c, R), sreg(s0c, s1c, R + 1), c2, nmc, l0, l1, wc[c >> 2][c & 3], mxa, mxb,
                              s0n[q4], s0n[q4 + 1], s1n[q4], s1n[q4 + 1], s0n[q4 + 2], s0n[q4 + 3], s1n[q4 + 2], s1n[q4 + 3], vf0, pp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        const int c = 9 + 2 * j, R = 2 * c;
        tp_mfma<T, 0>(o1, vf1, pp, nm);
        if (j == 0 || j == 3) {
          tp_vec<T, PRESC, 0>(sreg(s0c, s1c, R), sreg(s0c, s1c, R + 1), c2, nmc, l0, l1, wc[c >> 2][c & 3], mxa, mxb,
                              0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, vf1, pp);
        } else {
          const int q4 = 8 * (j - 1) + 4;
          tp_vec<T, PRESC, 2>(sreg(s0c, s1c, R), sreg(s0c, s1c, R + 1), c2, nmc, l0, l1, wc[c >> 2][c & 3], mxa, mxb,
                              s0n[q4], s0n[q4 + 1], s1n[q4], s1n[q4 + 1], s0n[q4 + 2], s0n[q4 + 3], s1n[q4 + 2], s1n[q4 + 3], vf1, pp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      vf0 = vn0; vf1 = vn1;
    }
    // ---- iteration boundary: P V of tile i-1 is in O; tile i is exponentiated and summed; tile i+1 has its row max ----
    if (!(TP_ABL & 8)) {
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(o0), "+v"(o1));   // last P V MFMAs -> VALU on O below (rare paths)
    apply_alpha();                                           // the reference change decided one iteration ago
    if (FOLD) {
      if (fold_pend) fold_boundary(fold_seg, fold_next, ls_pend);   // the segment that closed with tile i-1
      fold_pend = false;
      if (cur_last || !more) {      // tile i closes its segment (or the piece): capture its row sum, fold after its P V
        ls_pend = row_sum();
        l0 = 0.f; l1 = 0.f;
        fold_pend = true; fold_seg = cseg; fold_next = more;
      }
    }
    if (more) {
      const float mx = cross_max(max3(mxa, mxb, mxb));   // the two chains covered all 32 registers in clusters 10-13
      decide(mx, s0n, s1n, false);
    }
    }   // TP_ABL & 8
    cseg = nseg_i; ct0 = nt0_i;
    if (!(TP_ABL & 1)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this iteration's two transfers were issued at its start
      __syncthreads();
    }
  };

  int i = 0;
  for (; i + 1 < NTILES; i += 2) {
    step(i, s0x, s1x, s0y, s1y, wx, wy);
    step(i + 1, s0y, s1y, s0x, s1x, wy, wx);
  }
  // ---- drain: P V of the last tile (its V tile was issued in the last iteration and landed before its barrier) -----------
  auto drain = [&](unsigned (&w)[4][4]) {
    const unsigned char* Vb = smem + V_OFF + ((NTILES - 1) & 1) * TILE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v8 pp = __builtin_bit_cast(v8, u32x4{w[j][0], w[j][1], w[j][2], w[j][3]});
      const v8 a0 = v_frag(Vb, 0, j), a1 = v_frag(Vb, 1, j);
      tp_mfma<T, 0>(o0, a0, pp, nm);
      tp_mfma<T, 0>(o1, a1, pp, nm);
    }
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(o0), "+v"(o1));
    apply_alpha();
    if (FOLD && fold_pend) fold_boundary(fold_seg, false, ls_pend);
  };
  // an even-indexed step writes its probabilities to wy, an odd-indexed one to wx
  if (i < NTILES) { step(i, s0x, s1x, s0y, s1y, wx, wy); drain(wy); }
  else drain(wx);

  // ---- epilogue -----------------------------------------------------------------------------------------------------
  float l_fin = FOLD ? l_done : row_sum();
  const float m_raw = PRESC ? m_run / p.scale_log2 : m_run;   // the combine kernel and the LSE work in raw-score units
  if (npiece > 1) {
    const int64_t prow = ((int64_t)((xcd * (p.sk_ix - p.sk_full) + (item_local - p.sk_full)) * npiece + piece)) * QB + wid * 32 + lq;
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
}

template <typename T, bool FOLD, bool PRESC>
hipError_t launch(const AttnKParams& p0, hipStream_t s) {
  AttnKParams p = p0;
  constexpr int QB = 256;
  p.nqb = (p.Lq + QB - 1) / QB;
  p.sk_items = p.B * p.H * p.nqb;
  p.sk_ix = (p.sk_items + 7) / 8;
  const int slots_x = 32;   // one 8-wave workgroup per CU, 32 CUs per XCD
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
  hipLaunchKernelGGL((shared_attn_fwd_tp_kernel<T, FOLD, PRESC>), dim3(grid), dim3(512), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || k <= 1) return e;
  return ir_launch_shared_attn_combine(p, std::is_same<T, __bf16>::value ? 1 : 0, QB, rem, s);
}

template <typename T>
hipError_t launch_t(const AttnKParams& p, hipStream_t s) {
  const bool fold = p.aa != nullptr;
  if (p.q_prescaled) return fold ? launch<T, true, true>(p, s) : launch<T, false, true>(p, s);
  return fold ? launch<T, true, false>(p, s) : launch<T, false, false>(p, s);
}

}  // namespace

hipError_t ir_launch_shared_attn_fwd_tp(const AttnKParams& p, int dtype, hipStream_t s) {
  return dtype == 1 ? launch_t<__bf16>(p, s) : launch_t<_Float16>(p, s);
}

#endif  // IR_ABLATIONS
