// shared_attn_fwd_w128.hip - the fused extended self-attention forward with ONE wave per SIMD and 128 query rows per wave
// (gfx950; round 6).  Same math, layouts and C-ABI contract as the other kernels (attn_processors.py:232-264 of the reference:
// softmax(scale Q [K_self? ; K_ref_0 ; ...]^T) [V_self? ; a_n V_ref_n + b_n ; ...]), pre-scaled Q only (IR_FLAG_Q_PRESCALED).
//
// Why: in the 64-row kernel (two waves per SIMD, compiler-scheduled) both waves of a SIMD leave the per-tile barrier in the
// same phase, so the softmax's vector work (64 v_exp_f32 + 64 adds + 32 conversions per wave and tile) runs beside no MFMA:
// matrix-pipe occupancy 44 %, VALU-active + MFMA-busy = 93 % of the SIMD cycles (profiles/r5_pmc_shared_attn.txt).  Here a wave
// owns the whole 512-register file and FOUR 32-row blocks; its instruction stream is a software pipeline over the blocks -
// while the matrix pipe works on QK^T of block i+1 and P.V of block i-1, the vector ALU exponentiates block i - written out
// instruction by instruction (csrc/w128/gen.py -> shared_attn_fwd_w128_loop.inc: five single-issue vector instructions behind
// every MFMA, K / V^T fragments read from LDS once per tile for all four blocks into accumulator registers, LDS-DMA of the
// next tile inside the gaps, one barrier per tile, no scalar bookkeeping in the loop beyond a counter).
//
// Division of labour: the asm statement runs n consecutive tiles of ONE K/V segment (a "run"); everything rare - work decode,
// segment descriptors, the AdaIN fold at a segment boundary, the epilogue - is C++ around it.  State that outlives a run lives
// in accumulator registers only (O a[0:127], Q fragments a[128:191], m / l / l_done a[192:203]); the compiler owns v0-v31 and
// v224-v255 and never touches an AGPR (tools/check_resources.py fails the build on a spill or scratch use of this kernel;
// tests/test_build_guards.py disassembles it and fails on any compiler-generated v_accvgpr_* / AGPR operand outside the asm blocks).
#include <type_traits>
#include <utility>

#include "ir_common.h"
#include "ir_kernels.h"
#include "shared_attn_fwd_w128_loop.inc"

namespace {

constexpr int KVB = IR_KV_TILE;
constexpr int TILE_BYTES = KVB * 64 * 2;  // 8 KiB
constexpr int NW = 4, QB = NW * 128;      // 4 waves x 128 rows

// accumulator registers by compile-time index (asm-owned state; see the header)
template <int I>
static __device__ __forceinline__ float acc_get() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(I));
  return x;
}
template <int I>
static __device__ __forceinline__ void acc_set(float x) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(x), "n"(I));
}
template <int I>
static __device__ __forceinline__ void acc_set_u(unsigned x) {
  asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(x), "n"(I));
}
template <int N, typename F, int... Is>
static __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F>
static __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl<N>(f, std::make_integer_sequence<int, N>{});
}

constexpr int A_O = 0, A_Q = 128, A_M = 192, A_L = 196, A_LD = 200;
constexpr int FOLD_MAX_SEG = 16;   // segment lists up to 16 entries take their fold coefficients from the LDS table (longer ones: from global memory)

template <typename T, bool FOLD>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 1))) shared_attn_fwd_w128_kernel(const AttnKParams p) {
  using v4 = typename ElemTraits<T>::v4;
  // K ring of 2, V ring of 2 (32 KiB, at LDS offset 0); FOLD: behind it the fold coefficients of every segment boundary
  // ([segment][another segment follows in this piece][a_cur / a_next | b_cur / a_next][64 channels] fp32, 1 KiB per segment)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TILE_BYTES + (FOLD ? FOLD_MAX_SEG * 1024 : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lq = lane & 31;

  // ---- work decode: whole items, then K/V-range pieces of the remainder items (as the 64-row kernel, 512-row items) --------
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
  const int bh = lin / p.nqb, qb = lin - bh * p.nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int tile_begin = (int)(((long)p.ntiles * piece) / npiece);
  const int tile_end = (int)(((long)p.ntiles * (piece + 1)) / npiece);
  const int NTILES = tile_end - tile_begin;

  // ---- lane constants of the run statement -----------------------------------------------------------------------------------
  const unsigned lds0 = (unsigned)(unsigned long long)(IR_LDS unsigned char*)smem;
  if (lds0 & 0x3fffu) __builtin_trap();   // the stream toggles ring slots with XOR 8192: the ring must sit on a 16-KiB boundary (it is the only LDS object: 0)
  unsigned ka[4], va[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ka[ks] = lds0 + lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);
  {
    const int m = lane & 15, g = (lane >> 4) & 1;
    const int sw = (m >> 3) & 1;
#pragma unroll
    for (int db = 0; db < 2; ++db) va[db] = lds0 + (4 * hi + (m >> 2)) * 128 + ((db ^ sw) << 6) + 32 * g + 8 * (m & 3);
  }
  const int wb = __builtin_amdgcn_readfirstlane((int)(lds0 + wid * 1024));
  const int pslot = tid & 7;
  int srow[2];
  srow[0] = tid >> 3;
  srow[1] = (tid >> 3) + 32;

  auto pair_sum = [&](float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  };

  // FOLD (AdaIN, ratio frame as in the 64-row kernel): the accumulators hold (sum over finished segments of (a_s o O_s + b_s l_s)
  // + a_cur o O_cur) / a_cur; closing segment `sc`: acc <- acc * (a_cur / a_next) + l_cur * (b_cur / a_next), a = 1, b = 0 for
  // the self segment and behind the last one
  auto fold_boundary = [&](int sc, bool has_next) {
    float ls[4];
    static_for<4>([&](auto blk) {
      constexpr int B_ = decltype(blk)::value;
      ls[B_] = pair_sum(acc_get<A_L + B_>());
      acc_set<A_LD + B_>(acc_get<A_LD + B_>() + ls[B_]);
      acc_set<A_L + B_>(0.f);
    });
    const bool cur_ref = !(p.include_self && sc == 0);
    const bool tab = (p.include_self + p.N) <= FOLD_MAX_SEG;
    const int64_t ao_c = ((int64_t)(b * p.N + (cur_ref ? sc - p.include_self : 0)) * p.H + h) * 64 + 4 * hi;
    const int64_t ao_n = ((int64_t)(b * p.N + (has_next ? sc + 1 - p.include_self : 0)) * p.H + h) * 64 + 4 * hi;
    const float* const trow = (const float*)(smem + 4 * TILE_BYTES) + ((sc * 2 + (has_next ? 1 : 0)) * 2) * 64 + 4 * hi;
    static_for<4>([&](auto g4_) {
      constexpr int g4 = decltype(g4_)::value;
      f32x4 s_a0, s_a1, s_b0, s_b1;      // a_cur / a_next and b_cur / a_next of this lane's channels 8 g4 + 4 hi .. + 3 (and + 32)
      if (tab) {
        // round 6: computed once per item into LDS by the prologue (same operations, same bits): no loads from global memory,
        // no divisions and no dependent wait per boundary - the fold was ~5 K cycles per boundary, 2 % of an item at the top layer
        s_a0 = *(const f32x4*)(trow + 8 * g4); s_a1 = *(const f32x4*)(trow + 32 + 8 * g4);
        s_b0 = *(const f32x4*)(trow + 64 + 8 * g4); s_b1 = *(const f32x4*)(trow + 96 + 8 * g4);
      } else {
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
          const float i0 = 1.0f / an0[i], i1 = 1.0f / an1[i];
          s_a0[i] = ac0[i] * i0; s_b0[i] = bc0[i] * i0; s_a1[i] = ac1[i] * i1; s_b1[i] = bc1[i] * i1;
        }
      }
      static_for<4>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int r = 4 * g4 + i;
        const float sa0 = s_a0[i], sb0 = s_b0[i], sa1 = s_a1[i], sb1 = s_b1[i];
        static_for<4>([&](auto blk) {
          constexpr int B_ = decltype(blk)::value;
          acc_set<A_O + 32 * B_ + r>(__builtin_fmaf(acc_get<A_O + 32 * B_ + r>(), sa0, ls[B_] * sb0));
          acc_set<A_O + 32 * B_ + 16 + r>(__builtin_fmaf(acc_get<A_O + 32 * B_ + 16 + r>(), sa1, ls[B_] * sb1));
        });
      });
    });
  };

  // ---- the K/V walk: runs of consecutive tiles of one segment ---------------------------------------------------------------
  // The DMA stream runs across runs (stream v2): while a run computes its last tile, its last DMA slot fetches the FIRST tile
  // of the next run (flags bit 1; the next segment's descriptors are operands of their own), and the next run starts with
  // that tile already on its way (flags bit 0).  Ring-slot parity continues from run to run: `gt` counts the piece's tiles.
  struct RunP {
    const T* sk;
    const T* sv;
    int ksl_b, vsl_b, slen, seg, t0, n;
  };
  auto run_params = [&](int t) -> RunP {
    RunP r;
    int seg_tiles;
    if (p.include_self && t < p.tiles_self) {
      r.seg = 0; r.t0 = t; seg_tiles = p.tiles_self;
      r.sk = (const T*)p.k_self + (int64_t)b * p.ks_sb + (int64_t)h * p.ks_sh;
      r.sv = (const T*)p.v_self + (int64_t)b * p.vs_sb + (int64_t)h * p.vs_sh;
      r.ksl_b = (int)p.ks_sl * 2; r.vsl_b = (int)p.vs_sl * 2; r.slen = p.Ls;
    } else {
      const int rr = t - p.tiles_self;
      const int n = rr / p.tiles_ref;
      r.seg = p.include_self + n; r.t0 = rr - n * p.tiles_ref; seg_tiles = p.tiles_ref;
      r.sk = (const T*)p.k_ref + (int64_t)b * p.kr_sb + (int64_t)n * p.kr_sn + (int64_t)h * p.kr_sh;
      r.sv = (const T*)p.v_ref + (int64_t)b * p.vr_sb + (int64_t)n * p.vr_sn + (int64_t)h * p.vr_sh;
      r.ksl_b = (int)p.kr_sl * 2; r.vsl_b = (int)p.vr_sl * 2; r.slen = p.Lr;
    }
    const int seg_left = seg_tiles - r.t0, piece_left = tile_end - t;
    r.n = seg_left < piece_left ? seg_left : piece_left;
    return r;
  };
  // The item's FIRST tile goes on its way before anything else (its descriptors need nothing but the work decode), so that its
  // latency overlaps the Q loads and the accumulator initialisation below; the first run then starts with flags bit 0 set.
  if (tile_begin < tile_end) {
    const RunP r0 = run_params(tile_begin);
    const i32x4 kd0 = make_rsrc_words(r0.sk, (unsigned)((r0.slen - 1) * r0.ksl_b + 128));
    const i32x4 vd0 = make_rsrc_words(r0.sv, (unsigned)((r0.slen - 1) * r0.vsl_b + 128));
    const int ksoff0 = r0.t0 * KVB * r0.ksl_b, vsoff0 = r0.t0 * KVB * r0.vsl_b;
    unsigned ko0[2], vo0[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      ko0[c] = (unsigned)(srow[c] * r0.ksl_b + ((pslot ^ ((srow[c] >> 1) & 7)) * 16));
      vo0[c] = (unsigned)(srow[c] * r0.vsl_b + ((pslot ^ (((srow[c] >> 1) & 1) << 2)) * 16));
    }
    asm volatile("s_nop 4\n\t" W128_DMA0_ASM
                 :
                 : [ko0] "v"(ko0[0]), [ko1] "v"(ko0[1]), [vo0] "v"(vo0[0]), [vo1] "v"(vo0[1]), [kd] "s"(kd0), [vd] "s"(vd0),
                   [ksoff] "s"(ksoff0), [vsoff] "s"(vsoff0), [wb] "s"(wb)
                 : "memory");
  }

  if (FOLD && (p.include_self + p.N) <= FOLD_MAX_SEG) {
    // fold coefficients of every segment boundary, both forms (another segment follows in this piece / the piece or the walk ends
    // here: a_next = 1): one entry per thread and step, coalesced reads of the affine; visible to every wave behind the first
    // run statement's barrier
    float* const tabw = (float*)(smem + 4 * TILE_BYTES);
    const int nseg_all = p.include_self + p.N;
    for (int e = tid; e < nseg_all * 128; e += NW * 64) {
      const int sg = e >> 7, var = (e >> 6) & 1, d = e & 63;
      const bool cref = !(p.include_self && sg == 0);
      const bool nxt = var == 1 && sg + 1 < nseg_all;
      const int64_t oc = ((int64_t)(b * p.N + (cref ? sg - p.include_self : 0)) * p.H + h) * 64 + d;
      const int64_t on = ((int64_t)(b * p.N + (nxt ? sg + 1 - p.include_self : 0)) * p.H + h) * 64 + d;
      const float ac = cref ? p.aa[oc] : 1.f, bc = cref ? p.ab[oc] : 0.f, an = nxt ? p.aa[on] : 1.f;
      const float inv = 1.0f / an;
      tabw[((sg * 2 + var) * 2) * 64 + d] = ac * inv;
      tabw[((sg * 2 + var) * 2 + 1) * 64 + d] = bc * inv;
    }
  }

  // ---- state: O = 0, m = l = l_done = 0; Q fragments of the four row blocks into a[128:191] ---------------------------------
  static_for<128>([&](auto i) { acc_set<A_O + decltype(i)::value>(0.f); });
  static_for<12>([&](auto i) { acc_set<A_M + decltype(i)::value>(0.f); });
  const int qrow0 = qb * QB + wid * 128 + lq;
  {
    const T* base = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)h * p.q_sh + hi * 8;
    static_for<4>([&](auto blk) {
      constexpr int B_ = decltype(blk)::value;
      const int row = qrow0 + 32 * B_;
      const int rc = row < p.Lq ? row : p.Lq - 1;
      static_for<4>([&](auto ks) {
        constexpr int KS = decltype(ks)::value;
        const u32x4 w = *(const u32x4*)(base + (int64_t)rc * p.q_sl + KS * 16);
        acc_set_u<A_Q + 16 * B_ + 4 * KS + 0>(w[0]);
        acc_set_u<A_Q + 16 * B_ + 4 * KS + 1>(w[1]);
        acc_set_u<A_Q + 16 * B_ + 4 * KS + 2>(w[2]);
        acc_set_u<A_Q + 16 * B_ + 4 * KS + 3>(w[3]);
      });
    });
  }

  if (tile_begin < tile_end) {
    int t = tile_begin, gt = 0;
    int first = 1, prefetched = 1;   // the first tile was issued above
    RunP cur = run_params(t);
    while (t < tile_end) {
      const int has_next = (t + cur.n < tile_end) ? 1 : 0;
      const RunP nx = has_next ? run_params(t + cur.n) : cur;
      const i32x4 kd = make_rsrc_words(cur.sk, (unsigned)((cur.slen - 1) * cur.ksl_b + 128));
      const i32x4 vd = make_rsrc_words(cur.sv, (unsigned)((cur.slen - 1) * cur.vsl_b + 128));
      const i32x4 nkd = make_rsrc_words(nx.sk, (unsigned)((nx.slen - 1) * nx.ksl_b + 128));
      const i32x4 nvd = make_rsrc_words(nx.sv, (unsigned)((nx.slen - 1) * nx.vsl_b + 128));
      const int kstep = KVB * cur.ksl_b, vstep = KVB * cur.vsl_b;
      const int ksoff = (cur.t0 + prefetched) * kstep, vsoff = (cur.t0 + prefetched) * vstep;   // the first tile THIS run fetches
      const int nksoff = nx.t0 * KVB * nx.ksl_b, nvsoff = nx.t0 * KVB * nx.vsl_b;
      unsigned ko[2], vo[2], nko[2], nvo[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        ko[c] = (unsigned)(srow[c] * cur.ksl_b + ((pslot ^ ((srow[c] >> 1) & 7)) * 16));
        vo[c] = (unsigned)(srow[c] * cur.vsl_b + ((pslot ^ (((srow[c] >> 1) & 1) << 2)) * 16));
        nko[c] = (unsigned)(srow[c] * nx.ksl_b + ((pslot ^ ((srow[c] >> 1) & 7)) * 16));
        nvo[c] = (unsigned)(srow[c] * nx.vsl_b + ((pslot ^ (((srow[c] >> 1) & 1) << 2)) * 16));
      }
      const unsigned par = (unsigned)(gt & 1) << 13;                       // ring slot of this run's first tile
      const int wbr = wb ^ (int)(((unsigned)((gt ^ prefetched) & 1)) << 13);   // ... of the first tile this run FETCHES
      const int thr = first ? (int)0xBF800000 : (int)0x45000000;   // -1.0: the item's first tile always takes the exact path; 2^11 after
      const int flags = prefetched | (has_next << 1);
#define W128_OPERANDS                                                                                                            \
  [ka0] "v"(ka[0] ^ par), [ka1] "v"(ka[1] ^ par), [ka2] "v"(ka[2] ^ par), [ka3] "v"(ka[3] ^ par), [va0] "v"(va[0] ^ par),        \
      [va1] "v"(va[1] ^ par), [ko0] "v"(ko[0]), [ko1] "v"(ko[1]), [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [nko0] "v"(nko[0]),        \
      [nko1] "v"(nko[1]), [nvo0] "v"(nvo[0]), [nvo1] "v"(nvo[1]), [kd] "s"(kd), [vd] "s"(vd), [nkd] "s"(nkd), [nvd] "s"(nvd),    \
      [kstep] "s"(kstep), [vstep] "s"(vstep), [ksoff] "s"(ksoff), [vsoff] "s"(vsoff), [nksoff] "s"(nksoff), [nvsoff] "s"(nvsoff), \
      [n] "s"(cur.n), [thr] "s"(thr), [wb] "s"(wbr), [flags] "s"(flags)
      if (std::is_same<T, __bf16>::value) {
        asm volatile(W128_RUN_ASM_BF16 : : W128_OPERANDS : W128_RUN_CLOBBERS);
      } else {
        asm volatile(W128_RUN_ASM_F16 : : W128_OPERANDS : W128_RUN_CLOBBERS);
      }
#undef W128_OPERANDS
      first = 0;
      t += cur.n;
      gt += cur.n;
      prefetched = has_next;
      if (FOLD) fold_boundary(cur.seg, t < tile_end);   // a piece that stops inside a segment closes what it has
      cur = nx;
    }
  }

  // ---- epilogue (per row block), as the 64-row kernel's --------------------------------------------------------------------------
  static_for<4>([&](auto blk) {
    constexpr int B_ = decltype(blk)::value;
    const int qrow = qrow0 + 32 * B_;
    const float l_fin = FOLD ? acc_get<A_LD + B_>() : pair_sum(acc_get<A_L + B_>());
    const float m_run = acc_get<A_M + B_>();
    const float m_raw = m_run / p.scale_log2;   // the combine kernel and the LSE work in raw-score units
    f32x16 o0, o1;
    static_for<16>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      o0[r] = acc_get<A_O + 32 * B_ + r>();
      o1[r] = acc_get<A_O + 32 * B_ + 16 + r>();
    });
    if (npiece > 1) {
      const int64_t prow = ((int64_t)((xcd * (p.sk_ix - p.sk_full) + (item_local - p.sk_full)) * npiece + piece)) * QB + wid * 128 + 32 * B_ + lq;
      float* wo = p.ws_o + prow * 64;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 x0, x1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { x0[i] = o0[4 * g4 + i]; x1[i] = o1[4 * g4 + i]; }
        *(f32x4*)(wo + 8 * g4 + 4 * hi) = x0;
        *(f32x4*)(wo + 32 + 8 * g4 + 4 * hi) = x1;
      }
      if (hi == 0) {
        p.ws_ml[prow * 2] = NTILES == 0 ? -INFINITY : m_raw;   // an empty piece stays out of the merge's maximum
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
        for (int i = 0; i < 4; ++i) { x0[i] = o0[4 * g4 + i] * inv; x1[i] = o1[4 * g4 + i] * inv; }
        if (p.out_f32) {
          *(f32x4*)(of + 8 * g4 + 4 * hi) = x0;
          *(f32x4*)(of + 32 + 8 * g4 + 4 * hi) = x1;
        } else {
          *(v4*)(op + 8 * g4 + 4 * hi) = __builtin_convertvector(x0, v4);
          *(v4*)(op + 32 + 8 * g4 + 4 * hi) = __builtin_convertvector(x1, v4);
        }
      }
      if (p.lse != nullptr && hi == 0)
        p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_raw * p.scale + __logf(l_fin);
    }
  });
}

template <typename T, bool FOLD>
hipError_t launch(const AttnKParams& p0, hipStream_t s) {
  AttnKParams p = p0;
  p.nqb = (p.Lq + QB - 1) / QB;
  p.sk_items = p.B * p.H * p.nqb;
  p.sk_ix = (p.sk_items + 7) / 8;
  const int slots_x = 32;   // one workgroup (four waves, one per SIMD) per CU, 32 CUs per XCD
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
  p.ws_cum = nullptr;
  const int grid = 8 * (full + rem * k);
  hipLaunchKernelGGL((shared_attn_fwd_w128_kernel<T, FOLD>), dim3(grid), dim3(NW * 64), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || k <= 1) return e;
  return ir_launch_shared_attn_combine(p, std::is_same<T, __bf16>::value ? 1 : 0, QB, rem, s);
}

}  // namespace

// What this kernel takes (everything else stays with the 64-row kernel): pre-scaled Q, whole 64-key tiles in every segment,
// no valid_refs closed form, no seg_mass by-product
bool ir_attn_w128_supports(const AttnKParams& p) {
  if (!p.q_prescaled || p.valid != nullptr || p.seg_cum != nullptr) return false;
  if (p.include_self && (p.Ls % KVB) != 0) return false;
  if (p.N > 0 && (p.Lr % KVB) != 0) return false;
  return p.ntiles > 0;
}

hipError_t ir_launch_shared_attn_fwd_w128(const AttnKParams& p, int dtype, hipStream_t s) {
  if (!ir_attn_w128_supports(p)) return hipErrorInvalidValue;
  if (p.aa != nullptr) return dtype == 1 ? launch<__bf16, true>(p, s) : launch<_Float16, true>(p, s);
  return dtype == 1 ? launch<__bf16, false>(p, s) : launch<_Float16, false>(p, s);
}
