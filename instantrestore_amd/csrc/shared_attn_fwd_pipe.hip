// shared_attn_fwd_pipe.hip - software-pipelined variant of the fused extended self-attention
// forward (gfx950).  Same math, layouts and C-ABI contract as shared_attn_fwd.hip; what changes
// is the schedule inside a wave.
//
// PMC + ablation of the straight-line kernel (profiles/, tools/gpu_ablate.py) showed that its
// phases - LDS fragment reads, QK^T MFMAs, softmax VALU, PV MFMAs, staging - are serialised per
// wave (removing any one buys ~20 %, additively) and that two waves per SIMD only overlap
// statistically.  Here each iteration t issues, in ONE basic block,
//
//        S(t+1) = K[t+1] Q^T          (8 MFMA, matrix pipe)
//        P(t)   = softmax step on S(t) (VALU: v_max3 / v_pk_fma / v_exp / v_cvt_pk)
//        O     += V[t]^T P(t)^T        (8 MFMA)
//
// so the matrix pipe works on tile t+1's scores while the VALU exponentiates tile t's.  K is
// staged one tile further ahead than V: K ring of 2, V ring of 3 (one prefetch stream, tiles
// arrive as (K[j], V[j]) pairs two iterations ahead of their PV).
//
// VALU diet (D = 64 is VALU-issue bound): row sums as packed v_pk_add_f32 into two 2-wide
// accumulators; scale-and-subtract as v_pk_fma_f32; AdaIN folded per SEGMENT with a lazily
// rescaled total that lives in LDS (private slot per thread), so the per-tile rescale touches
// only the 32 current accumulators and the fold costs nothing per tile.
#include <type_traits>

#include <stdlib.h>

#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int KVB = IR_KV_TILE;
constexpr int TILE_BYTES = KVB * 64 * 2;  // 8 KiB

// ABL (timing experiments only, WRONG results): 1 = no barrier, 2 = no LDS staging writes,
// 4 = no global loads, 8 = no exp/pack VALU, 16 = no LDS fragment reads
// MASS (ABI v9 seg_mass): also store the cumulative log-sum-exp at every segment boundary - a separate instantiation of the two
// default forms, launched only when the caller asks for the masses
template <typename T, int NW, bool FOLD, int ABL = 0, bool MASS = false>
__global__ void __launch_bounds__(NW * 64, ((ABL & 256) && !FOLD) ? 3 : 2) shared_attn_fwd_pipe_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CH = (KVB * 8) / NT;
  // CHK (pre-scaled Q only): no row max on ordinary tiles - the reference is checked AFTER the exponentials (below) and a
  // tile that outgrew it is formed again from its K tile, which therefore stays in LDS one step longer: K ring of 3
  constexpr bool CHK = (ABL & 8192) != 0;
  static_assert(!CHK || (ABL & 2048), "the check after the exponentials needs scores that already carry the reference");
  constexpr int KRING = CHK ? 3 : 2;
  constexpr int K_OFF = 0;                       // K ring: 2 tiles (3 with CHK)
  constexpr int V_OFF = KRING * TILE_BYTES;      // V ring: 3 tiles
  constexpr int OT_OFF = (KRING + 3) * TILE_BYTES;   // folded total: 32 floats per thread
  constexpr int LDS_BYTES = OT_OFF + (FOLD ? NT * 32 * 4 : 0);

  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int lq = lane & 31;

  // ---- work decode: XCD x = blockIdx % 8 owns items [x*sk_ix, (x+1)*sk_ix); its first sk_full
  // slots are whole items, the rest are K/V-range pieces of the remainder items (so the last,
  // partially filled round of workgroup slots is cut short instead of running at full length) --
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
  if (item_local >= p.sk_ix || lin >= p.sk_items) return;  // padding of the last XCD chunk
  const int bh = lin / p.nqb;
  const int qb = lin - bh * p.nqb;
  const int b = bh / p.H;
  const int h = bh - b * p.H;
  // ABI v8 (valid_refs): the all-zero suffix of the reference list (n >= valid[b]) is not walked; the piece that owns the end
  // of the range adds it in closed form before the epilogue (shared_attn_fwd_w64.hip has the derivation)
  int nref = p.N;
  if (p.valid != nullptr) {
    const int vb = p.valid[b];
    nref = vb < 0 ? 0 : (vb < p.N ? vb : p.N);
  }
  const int ntiles_b = p.tiles_self + nref * p.tiles_ref;
  const int tile_begin = (int)(((long)ntiles_b * piece) / npiece);
  const int tile_end = (int)(((long)ntiles_b * (piece + 1)) / npiece);

  const int qrow = qb * QB + wid * 32 + lq;
  const int qrow_c = qrow < p.Lq ? qrow : p.Lq - 1;
  v8 qf[4];
  {
    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)qrow_c * p.q_sl + (int64_t)h * p.q_sh + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8*)(qp + ks * 16);
  }
  // PRESC: Q carries scale*log2(e) (one more rounding of Q to the 16-bit type), so the scores leave the
  // matrix pipe already in the exp2 domain and - with the running reference fed through the C operand
  // of the first QK^T MFMA - already shifted: no per-score scale-and-subtract on the VALU.
  constexpr bool PRESC = (ABL & 2048) != 0;
  if (PRESC && !p.q_prescaled) {   // IR_FLAG_Q_PRESCALED: the projection already applied the factor (one rounding in all)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      qf[ks] = __builtin_convertvector(__builtin_convertvector(qf[ks], f32x8) * p.scale_log2, v8);
  }

  // ---- staging coordinates -------------------------------------------------------------------
  const int slot = tid & 7;
  int srow[CH], koff[CH], voff[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int row = (tid >> 3) + c * (NT / 8);
    srow[c] = row;
    koff[c] = row * 128 + ((slot ^ ((row >> 1) & 7)) << 4);
    voff[c] = row * 128 + ((slot ^ (((row >> 1) & 1) << 2)) << 4);
  }
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
  float* const ot_lds = (float*)(smem + OT_OFF) + tid;  // element r at ot_lds[r * NT]
  if (FOLD) {
#pragma unroll
    for (int r = 0; r < 32; ++r) ot_lds[r * NT] = 0.f;
  }

  // ---- prefetch stream (tile pairs j = 0, 1, 2, ...) ---------------------------------------------
  constexpr bool DMA_FLAG = (ABL & (64 | 128 | 256)) != 0;
  constexpr bool DMA_ASM = (ABL & (128 | 256)) != 0;  // DMA issued from inline asm, waited for by hand
  i32x4 krw = {0, 0, 0, 0}, vrw = {0, 0, 0, 0};
  const int nseg = p.include_self + nref;
  __amdgpu_buffer_rsrc_t krs, vrs;
  int kstep = 0, vstep = 0, sntile = 0;
  unsigned kvo[CH], vvo[CH];
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
    krs = __builtin_amdgcn_make_buffer_rsrc((void*)sk, 0, (slen - 1) * ksl_b + 128, 0x00020000);
    vrs = __builtin_amdgcn_make_buffer_rsrc((void*)sv, 0, (slen - 1) * vsl_b + 128, 0x00020000);
    if (DMA_ASM) {
      krw = make_rsrc_words(sk, (unsigned)((slen - 1) * ksl_b + 128));
      vrw = make_rsrc_words(sv, (unsigned)((slen - 1) * vsl_b + 128));
    }
    kstep = KVB * ksl_b;
    vstep = KVB * vsl_b;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      // register staging: thread fetches logical slot `slot` and writes it to the swizzled LDS
      // position; LDS-DMA: the LDS position is lane-linear (physical slot = tid & 7), so the thread
      // fetches the logical slot that belongs there (the XOR swizzles are involutions)
      const int ks_ = DMA_FLAG ? (slot ^ ((srow[c] >> 1) & 7)) : slot;
      const int vs_ = DMA_FLAG ? (slot ^ (((srow[c] >> 1) & 1) << 2)) : slot;
      kvo[c] = (unsigned)(srow[c] * ksl_b + ks_ * 16);
      vvo[c] = (unsigned)(srow[c] * vsl_b + vs_ * 16);
    }
  };
  u32x4 kreg[CH], vreg[CH];
  auto issue_dma = [&](int kslot, int vslot) {  // tile pair straight into its ring slots
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (DMA_ASM) {
        buffer_load_lds16_async(krw, smem + K_OFF + kslot * TILE_BYTES + (c * NW + wid) * 1024, kvo[c]);
        buffer_load_lds16_async(vrw, smem + V_OFF + vslot * TILE_BYTES + (c * NW + wid) * 1024, vvo[c]);
      } else {
        buffer_load_lds16(krs, smem + K_OFF + kslot * TILE_BYTES + (c * NW + wid) * 1024, kvo[c]);
        buffer_load_lds16(vrs, smem + V_OFF + vslot * TILE_BYTES + (c * NW + wid) * 1024, vvo[c]);
      }
      kvo[c] += kstep;
      vvo[c] += vstep;
    }
  };
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
  auto stage_write = [&](int kslot, int vslot) {
    unsigned char* Kb = smem + K_OFF + kslot * TILE_BYTES;
    unsigned char* Vb = smem + V_OFF + vslot * TILE_BYTES;
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

  // ---- state ---------------------------------------------------------------------------------
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  f32x2 la = {0.f, 0.f}, lb = {0.f, 0.f};  // this lane's partial row sums of the current segment
  float l_tot = 0.f;                        // FOLD: folded row sum, relative to m_ot
  float m_ot = -INFINITY;                   // FOLD: the running max the LDS total is scaled to
  float m_run = PRESC ? 0.f : -INFINITY;    // PRESC: the reference lives in the exp2 domain and starts at 0
  const float c2 = p.scale_log2;
  const float lazy_thr = PRESC ? 6.0f : 6.0f / c2;  // LAZYMAX: score growth that forces a rescale
  f32x16 mneg;                              // PRESC: -m_run in every register: C operand of the first QK^T MFMA
#pragma unroll
  for (int r = 0; r < 16; ++r) mneg[r] = 0.f;
  const int NTILES = tile_end - tile_begin;  // tiles of THIS piece (all of them when not split)

  // PV/softmax stream position (the QK^T of tile t+1 is issued unmasked; masking happens when a
  // tile reaches its softmax step, so the MFMAs and the VALU below share one basic block)
  int seg_b = 0, t0_b = tile_begin;  // (segment, tile within segment) of the first tile
  if (!(p.include_self && tile_begin < p.tiles_self)) {
    const int r = tile_begin - p.tiles_self;  // tiles_self == 0 without a self segment
    seg_b = p.include_self + r / p.tiles_ref;
    t0_b = r - (r / p.tiles_ref) * p.tiles_ref;
  }
  int cseg = seg_b, ct0 = t0_b;
  const bool first_is_self = (p.include_self && seg_b == 0);
  int c_ntile = first_is_self ? p.tiles_self : p.tiles_ref;
  int c_len = first_is_self ? p.Ls : p.Lr;

  constexpr bool NOPIPE = (ABL & 256) != 0;  // no S double buffer: fewer VGPRs, three waves per SIMD (implies asm DMA)
  constexpr bool DMA = (ABL & (64 | 128 | 256)) != 0;    // global->LDS staging by LDS-DMA (buffer_load ... lds), no registers (real variant)
  auto load_kf = [&](v8 (&kf0)[4], v8 (&kf1)[4], int kslot) {
    const unsigned char* Kb = smem + K_OFF + kslot * TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf0[ks] = (ABL & 16) ? qf[(ks + 1) & 3] : *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + kread[ks]);
      kf1[ks] = (ABL & 16) ? qf[(ks + 2) & 3] : *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + 32 * 128 + kread[ks]);
    }
  };
  auto qk_mfma = [&](f32x16& s0, f32x16& s1, const v8 (&kf0)[4], const v8 (&kf1)[4]) {
    if (PRESC) {
      s0 = Tr::mfma(kf0[0], qf[0], mneg);
      s1 = Tr::mfma(kf1[0], qf[0], mneg);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks) {
        s0 = Tr::mfma(kf0[ks], qf[ks], s0);
        s1 = Tr::mfma(kf1[ks], qf[ks], s1);
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s0 = Tr::mfma(kf0[ks], qf[ks], s0);
      s1 = Tr::mfma(kf1[ks], qf[ks], s1);
    }
  };
  auto qk = [&](f32x16& s0, f32x16& s1, int kslot) {
    v8 kf0[4], kf1[4];
    load_kf(kf0, kf1, kslot);
    qk_mfma(s0, s1, kf0, kf1);
  };
  // ABI v9 (seg_mass): cumulative log-sum-exp through segment s of this lane pair's row (log2 units) -> seg_cum, or this
  // piece's slot of ws_cum (shared_attn_fwd_w64.hip has the scheme).  The parameters are read from the kernel-argument segment
  // through a pointer the compiler cannot see through, so nothing of this stays in registers across the tile loop.
  typedef const __attribute__((address_space(4))) AttnKParams* KArgs;
  auto cold = [&]() -> KArgs { KArgs q = (KArgs)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(q)); return q; };
  auto cum_store = [&](int s, float v2) {
    const KArgs c = cold();
    if (hi != 0) return;
    if (npiece > 1) {
      const int64_t prow = ((int64_t)((xcd * (c->sk_ix - c->sk_full) + (item_local - c->sk_full)) * npiece + piece)) * QB + wid * 32 + lq;
      c->ws_cum[prow * c->nseg_out + s] = v2;
    } else if (qrow < c->Lq) {
      c->seg_cum[(((int64_t)b * c->H + h) * c->Lq + qrow) * c->nseg_out + s] = v2 * 0.69314718f;
    }
  };
  auto row_sum_now = [&]() {
    const float ls = (la[0] + la[1]) + (lb[0] + lb[1]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  };
  auto ref_log2 = [&]() { return PRESC ? m_run : m_run * c2; };
  // Fold the current segment's accumulators into the lazily scaled LDS total:
  //   total = total * 2^((m_ot - m_run) c) + O_seg o a_seg + rowsum(P_seg) * b_seg
  // (a = 1, b = 0 for the self segment).  Linear in the keys, so folding a PART of a segment (a
  // K/V-range piece ending mid-segment) is equally valid.
  auto fold_segment = [&]() {
    float lseg = (la[0] + la[1]) + (lb[0] + lb[1]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lseg), __float_as_uint(lseg), false, false);
      lseg = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float f = fast_exp2(PRESC ? (m_ot - m_run) : (m_ot - m_run) * c2);  // m_ot = -inf the first time: f = 0
    l_tot = l_tot * f + lseg;
    m_ot = m_run;
    if (MASS) cum_store(cseg, ref_log2() + __log2f(l_tot));   // l_tot: every segment so far, relative to m_run
    const bool is_ref = !(p.include_self && cseg == 0);
    const int n = cseg - p.include_self;
    const int64_t ao = ((int64_t)(b * p.N + (is_ref ? n : 0)) * p.H + h) * 64 + 4 * hi;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 a0 = {1.f, 1.f, 1.f, 1.f}, a1 = a0, b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (is_ref) {
        a0 = *(const f32x4*)(p.aa + ao + 8 * g4); a1 = *(const f32x4*)(p.aa + ao + 32 + 8 * g4);
        b0 = *(const f32x4*)(p.ab + ao + 8 * g4); b1 = *(const f32x4*)(p.ab + ao + 32 + 8 * g4);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        const float t0o = ot_lds[r * NT], t1o = ot_lds[(16 + r) * NT];
        ot_lds[r * NT] = __builtin_fmaf(o0[r], a0[i], __builtin_fmaf(lseg, b0[i], t0o * f));
        ot_lds[(16 + r) * NT] = __builtin_fmaf(o1[r], a1[i], __builtin_fmaf(lseg, b1[i], t1o * f));
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    la = f32x2{0.f, 0.f};
    lb = f32x2{0.f, 0.f};
  };

  // One pipeline step for tile t.  FAST = steady state (tiles t+1 and t+2 exist): no uniform
  // branches between the QK^T MFMAs of tile t+1 and the exp/PV work of tile t.
  auto step = [&](auto fast_tag, int t, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1, int vcur) {
    constexpr bool FAST = decltype(fast_tag)::value;
    const bool has1 = FAST || (t + 1 < NTILES), has2 = FAST || (t + 2 < NTILES);
    const unsigned char* Vb = smem + V_OFF + vcur * TILE_BYTES;

    // K ring slots: tile t lives in slot t & 1 (ring of 2) or in the V ring's slot (ring of 3: both rings turn together)
    const int v_next = vcur == 2 ? 0 : vcur + 1, v_prev = (vcur >= 1) ? vcur - 1 : 2;
    const int ks_cur = CHK ? vcur : (t & 1), ks_next = CHK ? v_next : ((t + 1) & 1), ks_free = CHK ? v_prev : (t & 1);
    if (NOPIPE) {  // straight schedule: prefetch pair t+1, S(t) now
      if (has1) {
        issue_dma((t + 1) & 1, vcur == 2 ? 0 : vcur + 1);
        advance();
      }
      qk(c0, c1, t & 1);
      asm volatile("s_nop 7\n\ts_nop 4" : "+v"(c0), "+v"(c1));  // MFMA -> asm v_max3 hazard pad
    }
    // EARLYQK: the prefetch issue and S(t+1) = K[t+1] Q^T go FIRST, so the row-max chain and the rescale
    // test of S(t) - 40-odd dependent VALU operations that used to run with the matrix pipe idle - execute
    // beside the eight QK^T MFMAs (S is double-buffered: n0/n1 are not the registers read below)
    constexpr bool EARLYQK = (ABL & 4096) != 0 && !PRESC && !NOPIPE;
    if (EARLYQK) {
      if (has2) {
        if (DMA) issue_dma(ks_free, v_prev);
        else issue_loads();
      }
      if (has1) qk(n0, n1, ks_next);
    }
    // (1) ragged tail of a segment: mask keys past its end (wave-uniform branch, rare)
    const int valid = c_len - ct0 * KVB;
    if (valid < KVB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= valid) c0[r] = -INFINITY;
        if (key + 32 >= valid) c1[r] = -INFINITY;
      }
    }
    // (2) row max of S(t): two v_max3 chains + one cross-half exchange
    auto row_max = [&](const f32x16& a0, const f32x16& a1) {
      float mxa = max3(a0[0], a0[1], a0[2]);
      float mxb = max3(a1[0], a1[1], a1[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) {
        mxa = max3(mxa, a0[r], a0[r + 1]);
        mxb = max3(mxb, a1[r], a1[r + 1]);
      }
      float m = max3(mxa, mxb, max3(a0[15], a1[15], a1[15]));
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
      return max3(__uint_as_float(sw[0]), __uint_as_float(sw[1]), m);
    };
    float mx = 0.f;
    if (!CHK) mx = row_max(c0, c1);
    if (PRESC && !CHK) {
      // S(t) is already c2*s - m_run.  First tile of the piece: adopt its row max whatever its sign (the
      // reference started at 0); later tiles: move only when some row grew by more than 2^6 (lazy).
      if (t == 0 || __any(mx > lazy_thr)) {
        const float d = t == 0 ? mx : (mx > 0.f ? mx : 0.f);
        const float alpha = t == 0 ? 1.f : fast_exp2(-d);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; c0[r] -= d; c1[r] -= d; }
        la *= alpha;
        lb *= alpha;
        m_run += d;
#pragma unroll
        for (int r = 0; r < 16; ++r) mneg[r] = -m_run;
      }
    }
    const float m_new = max3(m_run, mx, mx);
    // (3) rescale only when some row's max moved (exact).  LAZYMAX (experiment): keep the old
    // reference while no row's max grew by more than 2^6 in the exp2 domain, so P <= 64.
    constexpr bool LAZYMAX = (ABL & 1024) != 0;
    if (PRESC) {   // (CHK: nothing here - the check comes after the exponentials)
    } else if (LAZYMAX ? __any(mx > m_run + lazy_thr) : __any(m_new != m_run)) {
      const float alpha = fast_exp2((m_run - m_new) * c2);
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      la *= alpha;
      lb *= alpha;
      m_run = m_new;
    }
    const float mc = (LAZYMAX ? m_run : m_new) * c2;
    // (4) the overlapped block: prefetch issue, S(t+1) on the matrix pipe, exp/pack on the VALU,
    //     PV(t) on the matrix pipe
    if (!EARLYQK && !NOPIPE && has2 && !(ABL & 4)) {
      if (DMA) issue_dma(ks_free, v_prev);  // K slot of tile t+2 (ring of 2: (t+2)&1; of 3: (vcur+2)%3), V slot (vcur+2)%3: free since the last barrier
      else issue_loads();
    }
    if (!EARLYQK && !NOPIPE && has1) qk(n0, n1, ks_next);
    const f32x2 cc = {c2, c2};
    const f32x2 nm = {-mc, -mc};
    f32x2 ta = {0.f, 0.f}, tb = {0.f, 0.f};   // CHK: this tile's own row sums (four 8-element partial sums per lane)
    auto exps = [&]() {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        f32x2 t0v = {c0[r], c0[r + 1]};
        f32x2 t1v = {c1[r], c1[r + 1]};
        if (!PRESC) {
          t0v = __builtin_elementwise_fma(t0v, cc, nm);
          t1v = __builtin_elementwise_fma(t1v, cc, nm);
        }
        t0v[0] = fast_exp2(t0v[0]); t0v[1] = fast_exp2(t0v[1]);
        t1v[0] = fast_exp2(t1v[0]); t1v[1] = fast_exp2(t1v[1]);
        if (CHK) { ta += t0v; tb += t1v; }
        else { la += t0v; lb += t1v; }
        c0[r] = t0v[0]; c0[r + 1] = t0v[1];
        c1[r] = t1v[0]; c1[r + 1] = t1v[1];
      }
    };
    if (!CHK && !(ABL & 8)) exps();
    if (CHK) {
      // Reference checked AFTER the exponentials (the 64-row kernel's rule, shared_attn_fwd_w64.hip): the scores left the
      // matrix pipe as exponents relative to the running reference, so P = exp2(S) needs no row max; what must be caught
      // is a tile that OUTGROWS the reference.  Its own row sums show that: an 8-element partial sum above 2^11 means a
      // probability above 2^8, inf/NaN an overflow.  Then - and always on the first tile of the walk, whose reference is
      // still 0 - the scores are formed AGAIN (K tile still in LDS, P.V has not run, nothing of this tile was
      // accumulated) and the exact path runs on them: row max, reference moved by d, accumulators, row sums and the
      // already-formed S(t+1) shifted by d.  P <= 2^11 is harmless in the 16-bit operands (bf16 range; fp16 max 65504)
      // and in the fp32 accumulators.
      bool redo = (t == 0);
      if (!redo) {
        if (!(ABL & 8)) exps();
        const float worst = max3(fmaxf(ta[0], ta[1]), tb[0], tb[1]);
        redo = __any(!(worst <= 2048.f));
        if (redo) {
          qk(c0, c1, ks_cur);
          asm volatile("s_nop 7\n\ts_nop 4" : "+v"(c0), "+v"(c1));   // MFMA -> asm v_max3 hazard pad
          if (valid < KVB) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
              if (key >= valid) c0[r] = -INFINITY;
              if (key + 32 >= valid) c1[r] = -INFINITY;
            }
          }
        }
      }
      if (redo) {
        const float mxr = row_max(c0, c1);
        const float d = t == 0 ? mxr : (mxr > 0.f ? mxr : 0.f);
        const float alpha = t == 0 ? 1.f : fast_exp2(-d);     // first tile: the accumulators are empty (and 0 * inf = NaN)
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; c0[r] -= d; c1[r] -= d; n0[r] -= d; n1[r] -= d; }
        la *= alpha;
        lb *= alpha;
        m_run += d;
#pragma unroll
        for (int r = 0; r < 16; ++r) mneg[r] = -m_run;
        ta = f32x2{0.f, 0.f};
        tb = f32x2{0.f, 0.f};
        exps();
      }
      la += ta;
      lb += tb;
    }
    v8 pk[2][2];
    pk[0][0] = __builtin_convertvector(__builtin_shufflevector(c0, c0, 0, 1, 2, 3, 4, 5, 6, 7), v8);
    pk[0][1] = __builtin_convertvector(__builtin_shufflevector(c0, c0, 8, 9, 10, 11, 12, 13, 14, 15), v8);
    pk[1][0] = __builtin_convertvector(__builtin_shufflevector(c1, c1, 0, 1, 2, 3, 4, 5, 6, 7), v8);
    pk[1][1] = __builtin_convertvector(__builtin_shufflevector(c1, c1, 8, 9, 10, 11, 12, 13, 14, 15), v8);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = (32 * kb + 16 * ks) * 128;
        if (ABL & 16) {
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
    // (5) segment boundary of this stream: fold the AdaIN affine into the LDS total
    if (++ct0 == c_ntile) {
      if (FOLD) fold_segment();
      else if (MASS) cum_store(cseg, ref_log2() + __log2f(row_sum_now()));   // no fold: the sums run over all segments
      ct0 = 0;
      ++cseg;
      c_ntile = p.tiles_ref;
      c_len = p.Lr;
    }
    // (6) land pair t+2 in LDS
    if (!NOPIPE && has2) {
      if (DMA) {}
      else if (!(ABL & 2)) stage_write(t & 1, (vcur >= 1) ? vcur - 1 : 2);  // K slot (t+2)&1, V slot (vcur+2)%3
      else if (!(ABL & 4)) {
#pragma unroll
        for (int c = 0; c < CH; ++c) asm volatile("" ::"v"(kreg[c]), "v"(vreg[c]));  // keep the loads alive
      }
      advance();
    }
    if (DMA_ASM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // our DMA of pair t+2 has landed
    if (!(ABL & 1)) __syncthreads();
  };

  // ---- prologue: pairs 0 and 1 into LDS, S(0) ---------------------------------------------------
  seg = seg_b;
  t0 = t0_b;
  seg_setup(seg);
#pragma unroll
  for (int c = 0; c < CH; ++c) { kvo[c] += (unsigned)(t0 * kstep); vvo[c] += (unsigned)(t0 * vstep); }
  if (DMA) {
    issue_dma(0, 0);
    advance();
    if (!NOPIPE && NTILES > 1) {
      issue_dma(1, 1);
      advance();
    }
    if (DMA_ASM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    issue_loads();
    stage_write(0, 0);
    advance();
    if (NTILES > 1) {
      issue_loads();
      stage_write(1, 1);
      advance();
    }
  }
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));  // retire the Q loads before the loop
  __syncthreads();

  f32x16 sa0, sa1, sb0, sb1;
  if (!NOPIPE) {
    qk(sa0, sa1, 0);
    // S(0) is read by inline-asm v_max3 right away in the first step: pad the MFMA->VALU hazard
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(sa0), "+v"(sa1));
  }

  int t = 0, vcur = 0;  // vcur = V ring slot of tile t
  auto next3 = [](int v) { return v == 2 ? 0 : v + 1; };
  const std::integral_constant<bool, true> fast{};
  const std::integral_constant<bool, false> slow{};
  if (NOPIPE) {
    for (; t < NTILES; ++t) {
      step(slow, t, sa0, sa1, sa0, sa1, vcur);
      vcur = next3(vcur);
    }
  }
  for (; t + 3 < NTILES; t += 2) {  // steady state: both tiles of the pair have t+2 < NTILES
    step(fast, t, sa0, sa1, sb0, sb1, vcur);
    vcur = next3(vcur);
    step(fast, t + 1, sb0, sb1, sa0, sa1, vcur);
    vcur = next3(vcur);
  }
  for (; t < NTILES; t += 2) {      // tail (and tiny problems): runtime-checked steps
    step(slow, t, sa0, sa1, sb0, sb1, vcur);
    vcur = next3(vcur);
    if (t + 1 < NTILES) {
      step(slow, t + 1, sb0, sb1, sa0, sa1, vcur);
      vcur = next3(vcur);
    }
  }

  // ---- epilogue -----------------------------------------------------------------------------
  // a piece that stops inside a segment folds what it has.  Not an EMPTY piece (valid_refs can leave an item fewer tiles than
  // the split planned pieces for): nothing is accumulated, both references are still -inf and their difference is not a number
  const bool mid = ct0 != 0 && NTILES > 0;
  if (FOLD && mid) fold_segment();
  constexpr bool want_cum = MASS;
  if (!FOLD && want_cum && mid) cum_store(cseg, ref_log2() + __log2f(row_sum_now()));
  const int s_next = cseg + (mid ? 1 : 0);        // first segment whose cumulative value this piece has not stored yet
  if (want_cum && npiece > 1)
    for (int s = 0; s < seg_b; ++s) cum_store(s, -INFINITY);   // segments before this piece's range
  float l_pre = -1.f, pz_seg = 0.f;               // zero suffix: row sum before it (final frame), weight of one zero segment
  if (nref < p.N && piece == npiece - 1) {
    // zero-filled references in closed form (ABI v8): (N - nref) * Lr keys that all score exactly 0 and carry a zero value
    // row (with the fold: the AdaIN shift b).  The reference moves up to 0 if it was below; everything accumulated is
    // rescaled once; the row sum takes cnt * 2^(-m) and the folded total that weight times the suffix's summed shifts.
    const float nz = (float)(p.N - nref);
    if (!PRESC && m_run == -INFINITY) m_run = 0.f;             // no tile walked at all
    const float e = PRESC ? -m_run : -m_run * c2;
    const float up = max3(e, 0.f, 0.f);
    const float alpha = fast_exp2(-up);
    m_run += PRESC ? up : up / c2;
    const float pz = fast_exp2(e - up) * (float)p.Lr;          // weight of ONE zero segment (Lr keys); bs below sums the segments' shifts
    pz_seg = pz;
    if (want_cum) {
      // (through an opaque copy: sharing l_tot * alpha with the update below would change how THAT contracts into a
      //  multiply-add, and with it the last bit of the result relative to the launch without the by-product)
      float lt = FOLD ? l_tot : row_sum_now();
      asm volatile("" : "+v"(lt));
      l_pre = lt * alpha;
    }
    if (FOLD) {
      l_tot = l_tot * alpha + pz * nz;
      m_ot = m_run;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0;
        for (int n = nref; n < p.N; ++n) {
          const int64_t ao = ((int64_t)(b * p.N + n) * p.H + h) * 64 + 4 * hi;
          bs0 += *(const f32x4*)(p.ab + ao + 8 * g4);
          bs1 += *(const f32x4*)(p.ab + ao + 32 + 8 * g4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * g4 + i;
          ot_lds[r * NT] = __builtin_fmaf(pz, bs0[i], ot_lds[r * NT] * alpha);
          ot_lds[(16 + r) * NT] = __builtin_fmaf(pz, bs1[i], ot_lds[(16 + r) * NT] * alpha);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      la *= alpha;
      lb *= alpha;
      if (hi == 0) la[0] += pz * nz;                           // the two lanes of a row add their partial sums below
    }
  }
  float l_fin;
  if (FOLD) {
    l_fin = l_tot;  // everything is folded and m_ot == m_run
  } else {
    float ls = (la[0] + la[1]) + (lb[0] + lb[1]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
    l_fin = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (want_cum) {
    // segments after the end of this piece's range: the piece's total - and, on the piece that owns the zero-filled suffix, the
    // row sum before the suffix plus j zero segments
    const KArgs c = cold();
    const float lp = l_pre < 0.f ? l_fin : l_pre, mref = ref_log2();
    const int sz = c->include_self + nref;
    for (int s = s_next; s < c->nseg_out; ++s) {
      const int j = s - sz + 1;
      cum_store(s, mref + __log2f(lp + pz_seg * (float)(j > 0 ? j : 0)));
    }
  }
  if (npiece > 1) {
    // partial result of a K/V-range piece: unnormalised O (fp32), raw max, row sum -> workspace
    const int64_t prow = ((int64_t)((xcd * (p.sk_ix - p.sk_full) + (item_local - p.sk_full)) * npiece + piece)) * QB + wid * 32 + lq;
    float* wo = p.ws_o + prow * 64;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 x0, x1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        x0[i] = FOLD ? ot_lds[r * NT] : o0[r];
        x1[i] = FOLD ? ot_lds[(16 + r) * NT] : o1[r];
      }
      *(f32x4*)(wo + 8 * g4 + 4 * hi) = x0;
      *(f32x4*)(wo + 32 + 8 * g4 + 4 * hi) = x1;
    }
    if (hi == 0) {
      // the combine kernel weighs pieces by raw-score maxima.  An EMPTY piece (valid_refs can leave fewer tiles than pieces were
      // planned for) that does not own the zero-filled suffix holds nothing: it must not take part in the merge's maximum - the
      // pre-scaled forms start their reference at 0, and a 0 beside real pieces whose maxima lie below -126 exponent units
      // would flush every weight to zero
      const bool empty_piece = NTILES == 0 && !(nref < p.N && piece == npiece - 1);
      p.ws_ml[prow * 2] = empty_piece ? -INFINITY : (PRESC ? m_run / c2 : m_run);
      p.ws_ml[prow * 2 + 1] = l_fin;
    }
    return;
  }
  const float inv = 1.0f / l_fin;
  if (qrow < p.Lq) {
    const int64_t ooff = (int64_t)b * p.o_sb + (int64_t)qrow * p.o_sl + (int64_t)h * p.o_sh;
    T* op = (T*)p.out + ooff;
    float* of = (float*)p.out + ooff;   // IR_FLAG_OUT_F32: the result before the 16-bit rounding (strides in fp32 elements)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 x0, x1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        x0[i] = (FOLD ? ot_lds[r * NT] : o0[r]) * inv;
        x1[i] = (FOLD ? ot_lds[(16 + r) * NT] : o1[r]) * inv;
      }
      if (p.out_f32) {
        *(f32x4*)(of + 8 * g4 + 4 * hi) = x0;
        *(f32x4*)(of + 32 + 8 * g4 + 4 * hi) = x1;
      } else {
        *(v4*)(op + 8 * g4 + 4 * hi) = __builtin_convertvector(x0, v4);
        *(v4*)(op + 32 + 8 * g4 + 4 * hi) = __builtin_convertvector(x1, v4);
      }
    }
    if (p.lse != nullptr && hi == 0)
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = (PRESC ? m_run * 0.69314718f : m_run * p.scale) + __logf(l_fin);
  }
}

// Merge the K/V-range pieces of the remainder items: one workgroup per remainder item, thread =
// (row, 32-channel half).  out = sum_j w_j O_j / sum_j w_j l_j with w_j = 2^((m_j - M) c).
template <typename T, int QB>
__global__ void __launch_bounds__(256) shared_attn_combine_kernel(const AttnKParams p) {
  // grid: (8 * remainder items per XCD, QB / 16); 256 threads = 16 rows x 16 four-channel groups, so a
  // wave reads four whole 256-byte partial rows per load and thousands of workgroups are in flight
  // (one workgroup per item left half the CUs idle on a latency-bound kernel)
  using v4 = typename ElemTraits<T>::v4;
  const int xcd = blockIdx.x & 7, ri = blockIdx.x >> 3;  // ri: remainder item index inside the XCD chunk
  const int rem_x = p.sk_ix - p.sk_full;
  const int item_local = p.sk_full + ri;
  const int lin = xcd * p.sk_ix + item_local;
  if (ri >= rem_x || lin >= p.sk_items) return;
  const int bh = lin / p.nqb, qb = lin - bh * p.nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int row = blockIdx.y * 16 + (threadIdx.x >> 4), grp = threadIdx.x & 15;
  const int qrow = qb * QB + row;
  if (qrow >= p.Lq) return;
  const int64_t base = (int64_t)(xcd * rem_x + ri) * p.sk_k;
  float M = -INFINITY;
  for (int j = 0; j < p.sk_k; ++j) M = fmaxf(M, p.ws_ml[((base + j) * QB + row) * 2]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float L = 0.f;
  for (int j = 0; j < p.sk_k; ++j) {
    const int64_t prow = (base + j) * QB + row;
    const float w = fast_exp2((p.ws_ml[prow * 2] - M) * p.scale_log2);
    L += w * p.ws_ml[prow * 2 + 1];
    const f32x4 x = *(const f32x4*)(p.ws_o + prow * 64 + grp * 4);
    acc += x * w;
  }
  const float inv = 1.0f / L;
  const int64_t off = (int64_t)b * p.o_sb + (int64_t)qrow * p.o_sl + (int64_t)h * p.o_sh + grp * 4;
  if (p.out_f32) *(f32x4*)((float*)p.out + off) = acc * inv;   // IR_FLAG_OUT_F32: the result before the 16-bit rounding
  else *(v4*)((T*)p.out + off) = __builtin_convertvector(acc * inv, v4);
  if (p.lse != nullptr && grp == 0) p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = M * p.scale + __logf(L);
  if (p.seg_cum != nullptr) {
    // ABI v9: a piece's entry s is the log-sum-exp (log2 units) of what ITS key range holds of segments 0 .. s (-inf before its
    // range, its total after it): the item's cumulative value is their log-sum-exp, taken in the fixed piece order
    for (int sg = grp; sg < p.nseg_out; sg += 16) {
      float mx = -INFINITY;
      for (int j = 0; j < p.sk_k; ++j) mx = fmaxf(mx, p.ws_cum[((base + j) * QB + row) * p.nseg_out + sg]);
      float sum = 0.f;
      for (int j = 0; j < p.sk_k; ++j) sum += fast_exp2(p.ws_cum[((base + j) * QB + row) * p.nseg_out + sg] - mx);
      p.seg_cum[(((int64_t)b * p.H + h) * p.Lq + qrow) * p.nseg_out + sg] = mx == -INFINITY ? -INFINITY : (mx + __log2f(sum)) * 0.69314718f;
    }
  }
}

// Work plan: items = B*H*ceil(Lq/QB); every XCD owns ix = ceil(items/8) consecutive items and
// has `slots_x` concurrently resident workgroups (CUs/8 * workgroups per CU).  Whole rounds run
// full K/V ranges; the remainder items are cut into k pieces so the last round ends early.
template <typename T, int NW, bool FOLD, int ABL = 0, bool MASS = false>
hipError_t launch(const AttnKParams& p0, hipStream_t s) {
  if (!MASS && p0.seg_cum != nullptr) return hipErrorInvalidValue;   // seg_mass: the two default forms carry the MASS instantiation
  AttnKParams p = p0;
  constexpr int QB = NW * 32;
  p.nqb = (p.Lq + QB - 1) / QB;
  p.sk_items = p.B * p.H * p.nqb;
  p.sk_ix = (p.sk_items + 7) / 8;
  const int slots_x = 32 * (NW == 8 ? 1 : (((ABL & 256) && !FOLD) ? 3 : 2));
  int full = (p.sk_ix / slots_x) * slots_x;
  int rem = p.sk_ix - full;
  int k = 1;
  if (p.ws != nullptr && rem > 0) {
    const size_t piece_bytes = (size_t)QB * (66 + (p.seg_cum != nullptr ? p.nseg_out : 0)) * sizeof(float);
    // pieces of at least 8 tiles.  (Round 3 tried 5-tile pieces for the 16x16-token class - 320 items of 20 tiles on 512
    // slots, cut in three - to put two workgroups on every CU: 47 us against 37 us unsplit, profiles/r3_layer_classes_cfg2_presc.txt:
    // the prologue, the fp32 partials and the combine launch cost more than the idle slots.)
    k = ir_pick_split(rem, slots_x, p.ntiles / 8, (long)(p.ws_bytes / piece_bytes / 8));
  }
  if (k <= 1) { full = p.sk_ix; rem = 0; k = 1; }
  {
    // measurement knob (round 6, VERDICT r5 item 5: "the split-K/V form on purpose"): IR_ATTN_FORCE_SPLIT=k cuts EVERY item of the
    // 32-row kernel's grid into k K/V-range pieces (more resident waves per CU to hide the per-tile latency chain), within the
    // workspace; read once, off by default.  profiles/r6_small_classes.txt has what it measures.
    static const int force_k = [] { const char* e = getenv("IR_ATTN_FORCE_SPLIT"); return e != nullptr ? atoi(e) : 0; }();
    if (force_k > 1 && p.ws != nullptr && p.ntiles >= 2 * force_k) {
      const size_t piece_bytes = (size_t)QB * (66 + (p.seg_cum != nullptr ? p.nseg_out : 0)) * sizeof(float);
      const long cap = (long)(p.ws_bytes / piece_bytes / 8);
      if ((long)p.sk_ix * force_k <= cap) { full = 0; rem = p.sk_ix; k = force_k; }
    }
  }
  p.sk_full = full;
  p.sk_k = k;
  p.ws_o = p.ws;
  p.ws_ml = p.ws + (size_t)8 * rem * k * QB * 64;
  p.ws_cum = p.ws_ml + (size_t)8 * rem * k * QB * 2;
  const int grid = 8 * (full + rem * k);
  hipLaunchKernelGGL((shared_attn_fwd_pipe_kernel<T, NW, FOLD, ABL, MASS>), dim3(grid), dim3(NW * 64), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || k <= 1) return e;
  hipLaunchKernelGGL((shared_attn_combine_kernel<T, QB>), dim3(8 * rem, QB / 16), dim3(256), 0, s, p);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_t(const AttnKParams& p, int nw, hipStream_t s) {
  const bool fold = (p.aa != nullptr);
  if (p.seg_cum != nullptr) {   // ABI v9 seg_mass: the two forms the default dispatch takes
    if (nw == 11) return fold ? launch<T, 4, true, 128 | 1024 | 2048, true>(p, s) : launch<T, 4, false, 128 | 1024 | 2048, true>(p, s);
    if (nw == 14) return fold ? launch<T, 4, true, 128 | 1024 | 4096, true>(p, s) : launch<T, 4, false, 128 | 1024 | 4096, true>(p, s);
    return hipErrorInvalidValue;
  }
#ifdef IR_ABLATIONS
  if (nw == 8) return fold ? launch<T, 8, true>(p, s) : launch<T, 8, false>(p, s);
  if (nw == 6) return fold ? launch<T, 4, true, 64>(p, s) : launch<T, 4, false, 64>(p, s);  // LDS-DMA staging
  if (nw == 9) return fold ? launch<T, 4, true, 256>(p, s) : launch<T, 4, false, 256>(p, s);  // straight schedule, 3 waves/SIMD
  if (nw == 4) return fold ? launch<T, 4, true>(p, s) : launch<T, 4, false>(p, s);             // register staging
#endif
  if (nw == 7) return fold ? launch<T, 4, true, 128>(p, s) : launch<T, 4, false, 128>(p, s);  // LDS-DMA from asm
  if (nw == 10) return fold ? launch<T, 4, true, 128 | 1024>(p, s) : launch<T, 4, false, 128 | 1024>(p, s);  // asm DMA + lazy max
  if (nw == 11) return fold ? launch<T, 4, true, 128 | 1024 | 2048>(p, s) : launch<T, 4, false, 128 | 1024 | 2048>(p, s);  // + pre-scaled Q, reference through the C operand
  if (nw == 14) return fold ? launch<T, 4, true, 128 | 1024 | 4096>(p, s) : launch<T, 4, false, 128 | 1024 | 4096>(p, s);  // + QK^T of the next tile first
  if (nw == 18) return fold ? launch<T, 4, true, 128 | 1024 | 2048 | 8192>(p, s) : launch<T, 4, false, 128 | 1024 | 2048 | 8192>(p, s);  // 11 + reference checked after the exponentials
  return hipErrorInvalidValue;
}

}  // namespace

// Ablation builds (timing experiments, WRONG results) exist only with -DIR_ABLATIONS
// (`build.sh -DIR_ABLATIONS`, used by tools/gpu_ablate.py); the shipped library has none.
hipError_t ir_launch_shared_attn_fwd_pipe_abl(const AttnKParams& p, int abl, hipStream_t s) {
#ifndef IR_ABLATIONS
  (void)abl;
  return launch<__bf16, 4, false, 0>(p, s);
#else
  switch (abl) {
    case 1: return launch<__bf16, 4, false, 1>(p, s);
    case 2: return launch<__bf16, 4, false, 2>(p, s);
    case 3: return launch<__bf16, 4, false, 3>(p, s);
    case 4: return launch<__bf16, 4, false, 4>(p, s);
    case 6: return launch<__bf16, 4, false, 6>(p, s);
    case 7: return launch<__bf16, 4, false, 7>(p, s);
    case 8: return launch<__bf16, 4, false, 8>(p, s);
    case 15: return launch<__bf16, 4, false, 15>(p, s);
    case 16: return launch<__bf16, 4, false, 16>(p, s);
    case 23: return launch<__bf16, 4, false, 23>(p, s);
    case 31: return launch<__bf16, 4, false, 31>(p, s);
    default: return launch<__bf16, 4, false, 0>(p, s);
  }
#endif
}

// combine launcher shared with the ping-pong kernel (p must carry the final sk_* / ws_* fields)
hipError_t ir_launch_shared_attn_combine(const AttnKParams& p, int dtype, int qb, int rem, hipStream_t s) {
  if (qb == 512) {
    if (dtype == 1) hipLaunchKernelGGL((shared_attn_combine_kernel<__bf16, 512>), dim3(8 * rem, 32), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((shared_attn_combine_kernel<_Float16, 512>), dim3(8 * rem, 32), dim3(256), 0, s, p);
  } else if (qb == 256) {
    if (dtype == 1) hipLaunchKernelGGL((shared_attn_combine_kernel<__bf16, 256>), dim3(8 * rem, 16), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((shared_attn_combine_kernel<_Float16, 256>), dim3(8 * rem, 16), dim3(256), 0, s, p);
  } else {
    if (dtype == 1) hipLaunchKernelGGL((shared_attn_combine_kernel<__bf16, 128>), dim3(8 * rem, 8), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((shared_attn_combine_kernel<_Float16, 128>), dim3(8 * rem, 8), dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

hipError_t ir_launch_shared_attn_fwd_pipe(const AttnKParams& p, int dtype, int nw, hipStream_t s) {
  return dtype == 1 ? launch_t<__bf16>(p, nw, s) : launch_t<_Float16>(p, nw, s);
}
