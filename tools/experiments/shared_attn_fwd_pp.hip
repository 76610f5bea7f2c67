// shared_attn_fwd_pp.hip - "ping-pong" variant of the fused extended self-attention forward
// (gfx950).  Same math, layouts and C-ABI contract as the other two attention kernels.
//
// Why: PMC on the pipelined kernel shows the SIMD's VALU/MFMA issue port 61 % busy and the
// matrix pipe 39 % busy - two uncoordinated waves per SIMD leave both units idle part of the
// time.  Here ONE 8-wave workgroup per CU runs its two wave groups (waves 0-3 = A, 4-7 = B; one
// wave of each per SIMD) half a step apart, separated by s_barrier:
//
//      half-step:   2t          2t+1         2t+2         2t+3
//      group A:     V(t)        M(t)         V(t+1)       M(t+1)
//      group B:     M(t-1)      V(t)         M(t)         V(t+1)
//
//   V(t) = softmax step of tile t: mask, v_max3 chains, rescale, exp/pack -> P(t)    (VALU only)
//   M(t) = O += V[t]^T P(t)^T ; S(t+1) = K[t+1] Q^T                                 (MFMA + LDS reads)
//
// so on every SIMD one wave feeds the matrix pipe while its partner occupies the VALU port.
// S is single-buffered (S(t+1) is written in M(t), consumed in V(t+1)); tiles arrive by
// asm-issued LDS-DMA as (K[j], V[j]) pairs into a ring of 3 slots: pair j is issued at half-step
// 2j-3 (after the last read of pair j-3) and waited for before the barrier closing half-step
// 2j-2, i.e. with two half-steps of flight time.  256 query rows per workgroup halves the
// global->LDS traffic and the barriers per flop relative to the 4-wave kernels.
#ifdef IR_ABLATIONS   // documented experiment (variant 8): development builds only
#include <type_traits>

#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int KVB = IR_KV_TILE;
constexpr int TILE_BYTES = KVB * 64 * 2;  // 8 KiB

template <typename T, bool FOLD>
__global__ void __launch_bounds__(512, 2) shared_attn_fwd_pp_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NT = 512, QB = 256;  // 8 waves
  constexpr int K_OFF = 0;                      // K ring: 3 tiles
  constexpr int V_OFF = 3 * TILE_BYTES;         // V ring: 3 tiles
  constexpr int OT_OFF = 6 * TILE_BYTES;        // folded total: 32 floats per thread
  constexpr int LDS_BYTES = OT_OFF + (FOLD ? NT * 32 * 4 : 0);
  __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool grpB = wid >= 4;  // wave-uniform
  const int hi = lane >> 5, lq = lane & 31;

  // ---- work decode (same plan as the pipelined kernel: whole items, then K/V-range pieces) ----
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

  const int qrow = qb * QB + wid * 32 + lq;
  const int qrow_c = qrow < p.Lq ? qrow : p.Lq - 1;
  v8 qf[4];
  {
    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)qrow_c * p.q_sl + (int64_t)h * p.q_sh + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8*)(qp + ks * 16);
  }

  // ---- LDS read offsets (identical swizzles to the other kernels) -------------------------------
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
  float* const ot_lds = (float*)(smem + OT_OFF) + tid;
  if (FOLD) {
#pragma unroll
    for (int r = 0; r < 32; ++r) ot_lds[r * NT] = 0.f;
  }

  // ---- DMA stream: 512 threads cover one 8-KiB tile with one 16-B chunk each ---------------------
  // chunk j = tid: row = tid >> 3, physical slot = tid & 7; the lane fetches the logical slot that
  // belongs at that physical position (XOR swizzles are involutions)
  const int srow = tid >> 3, pslot = tid & 7;
  const int nseg = p.include_self + p.N;
  i32x4 krw = {0, 0, 0, 0}, vrw = {0, 0, 0, 0};
  int kstep = 0, vstep = 0, sntile = 0;
  unsigned kvo = 0, vvo = 0;
  int seg = 0, t0 = 0;
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
    krw = make_rsrc_words(sk, (unsigned)((slen - 1) * ksl_b + 128));
    vrw = make_rsrc_words(sv, (unsigned)((slen - 1) * vsl_b + 128));
    kstep = KVB * ksl_b;
    vstep = KVB * vsl_b;
    kvo = (unsigned)(srow * ksl_b + ((pslot ^ ((srow >> 1) & 7)) * 16));
    vvo = (unsigned)(srow * vsl_b + ((pslot ^ (((srow >> 1) & 1) << 2)) * 16));
  };
  auto issue_pair = [&](int slot3) {  // this wave's 1 KiB of K[j] and of V[j] into ring slot `slot3`
    buffer_load_lds16_async(krw, smem + K_OFF + slot3 * TILE_BYTES + wid * 1024, kvo);
    buffer_load_lds16_async(vrw, smem + V_OFF + slot3 * TILE_BYTES + wid * 1024, vvo);
    kvo += kstep;
    vvo += vstep;
    if (++t0 == sntile) {
      t0 = 0;
      if (++seg < nseg) seg_setup(seg);
    }
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  // Raw barrier: __syncthreads() would also drain vmcnt(0) and kill the DMA that is meant to stay in
  // flight across it.  Every LDS read of a block is consumed (by an MFMA) inside the block, DMA
  // completion is waited for explicitly where the schedule needs it, the LDS total is thread-private.
  auto bar = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- state ------------------------------------------------------------------------------------
  f32x16 o0, o1, s0, s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s0[r] = 0.f; s1[r] = 0.f; }
  f32x2 la = {0.f, 0.f}, lb = {0.f, 0.f};
  float l_tot = 0.f, m_ot = -INFINITY, m_run = -INFINITY;
  const float c2 = p.scale_log2;
  v8 pk[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) pk[i][j][e] = (T)0.0f;

  int seg_b = 0, t0_b = tile_begin;
  if (!(p.include_self && tile_begin < p.tiles_self)) {
    const int r = tile_begin - p.tiles_self;
    seg_b = p.include_self + r / p.tiles_ref;
    t0_b = r - (r / p.tiles_ref) * p.tiles_ref;
  }
  int cseg = seg_b, ct0 = t0_b;
  const bool first_is_self = (p.include_self && seg_b == 0);
  int c_ntile = first_is_self ? p.tiles_self : p.tiles_ref;
  int c_len = first_is_self ? p.Ls : p.Lr;

  auto fold_segment = [&]() {
    float lseg = (la[0] + la[1]) + (lb[0] + lb[1]);
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lseg), __float_as_uint(lseg), false, false);
      lseg = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const float f = fast_exp2((m_ot - m_run) * c2);
    l_tot = l_tot * f + lseg;
    m_ot = m_run;
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

  // V(t): softmax step of tile t on (s0, s1) -> pk.  VALU only.
  auto v_block = [&]() {
    asm volatile("s_nop 7\n\ts_nop 4" : "+v"(s0), "+v"(s1));  // MFMA (end of the last M block) -> asm v_max3
    const int valid = c_len - ct0 * KVB;
    if (valid < KVB) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (key >= valid) s0[r] = -INFINITY;
        if (key + 32 >= valid) s1[r] = -INFINITY;
      }
    }
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
    const float m_new = max3(m_run, mx, mx);
    const float mc = m_new * c2;
    if (__any(m_new != m_run)) {
      const float alpha = fast_exp2(m_run * c2 - mc);
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      la *= alpha;
      lb *= alpha;
      m_run = m_new;
    }
    const f32x2 cc = {c2, c2};
    const f32x2 nm = {-mc, -mc};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 t0v = {s0[r], s0[r + 1]};
      f32x2 t1v = {s1[r], s1[r + 1]};
      t0v = __builtin_elementwise_fma(t0v, cc, nm);
      t1v = __builtin_elementwise_fma(t1v, cc, nm);
      t0v[0] = fast_exp2(t0v[0]); t0v[1] = fast_exp2(t0v[1]);
      t1v[0] = fast_exp2(t1v[0]); t1v[1] = fast_exp2(t1v[1]);
      la += t0v;
      lb += t1v;
      s0[r] = t0v[0]; s0[r + 1] = t0v[1];
      s1[r] = t1v[0]; s1[r + 1] = t1v[1];
    }
    pk[0][0] = __builtin_convertvector(__builtin_shufflevector(s0, s0, 0, 1, 2, 3, 4, 5, 6, 7), v8);
    pk[0][1] = __builtin_convertvector(__builtin_shufflevector(s0, s0, 8, 9, 10, 11, 12, 13, 14, 15), v8);
    pk[1][0] = __builtin_convertvector(__builtin_shufflevector(s1, s1, 0, 1, 2, 3, 4, 5, 6, 7), v8);
    pk[1][1] = __builtin_convertvector(__builtin_shufflevector(s1, s1, 8, 9, 10, 11, 12, 13, 14, 15), v8);
  };

  // M(t): O += V[t]^T P(t)^T (if do_pv), S(t+1) = K[t+1] Q^T (if do_qk).  MFMA + LDS reads only.
  auto m_block = [&](bool do_pv, int vslot, bool do_qk, int kslot) {
    if (do_qk) {
      const unsigned char* Kb = smem + K_OFF + kslot * TILE_BYTES;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const v8 a0 = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + kread[ks]);
        const v8 a1 = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Kb + 32 * 128 + kread[ks]);
        s0 = Tr::mfma(a0, qf[ks], s0);
        s1 = Tr::mfma(a1, qf[ks], s1);
      }
    }
    if (do_pv) {
      const unsigned char* Vb = smem + V_OFF + vslot * TILE_BYTES;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int off = (32 * kb + 16 * ks) * 128;
          const s16x4 a00 = lds_read_tr16(Vb + vread[0] + off);
          const s16x4 a01 = lds_read_tr16(Vb + vread[0] + off + 8 * 128);
          const s16x4 a10 = lds_read_tr16(Vb + vread[1] + off);
          const s16x4 a11 = lds_read_tr16(Vb + vread[1] + off + 8 * 128);
          o0 = Tr::mfma(join_tr<v8>(a00, a01), pk[kb][ks], o0);
          o1 = Tr::mfma(join_tr<v8>(a10, a11), pk[kb][ks], o1);
        }
      }
      if (++ct0 == c_ntile) {  // segment boundary: fold the AdaIN affine into the LDS total
        if (FOLD) fold_segment();
        ct0 = 0;
        ++cseg;
        c_ntile = p.tiles_ref;
        c_len = p.Lr;
      }
    }
  };

  auto next3 = [](int v) { return v == 2 ? 0 : v + 1; };

  // ---- prologue: pairs 0 and 1 land before anybody computes ------------------------------------
  seg = seg_b;
  t0 = t0_b;
  seg_setup(seg);
  kvo += (unsigned)(t0 * kstep);
  vvo += (unsigned)(t0 * vstep);
  issue_pair(0);
  if (NTILES > 1) issue_pair(1);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[ks]));
  dma_wait();
  __syncthreads();

  // Block sequence of every wave: M(-1), V(0), M(0), ..., V(T-1), M(T-1); group B runs it one
  // barrier later than group A.  Pair j (j >= 2) is issued by A at the start of M(j-2) and by B at
  // the start of V(j-2) - the same half-step - and waited for one barrier later.
  if (grpB) bar();                               // B's idle half-step
  m_block(false, 0, true, 0);                    // M(-1): S(0)
  bar();
  int slot_t = 0;                                // ring slot of pair t
  for (int t = 0; t < NTILES; ++t) {
    const int slot_t1 = next3(slot_t), slot_t2 = next3(slot_t1);
    const bool has1 = (t + 1 < NTILES), has2 = (t + 2 < NTILES);
    // ---- V(t) ----
    if (grpB && has2) issue_pair(slot_t2);       // B issues pair t+2 at the start of V(t)
    v_block();
    if (!grpB && t >= 1 && has1) dma_wait();     // A: pair t+1 (issued at the start of M(t-1)) has landed
    bar();
    // ---- M(t) ----
    if (!grpB && has2) issue_pair(slot_t2);      // A issues pair t+2 at the start of M(t)
    m_block(true, slot_t, has1, slot_t1);
    if (grpB && has2) dma_wait();                // B: pair t+2 (issued at the start of V(t)) has landed
    bar();
    slot_t = slot_t1;
  }
  if (!grpB) bar();                              // A's idle half-step

  // ---- epilogue (identical to the pipelined kernel) ------------------------------------------------
  if (FOLD && ct0 != 0) fold_segment();
  float l_fin;
  if (FOLD) {
    l_fin = l_tot;
  } else {
    float ls = (la[0] + la[1]) + (lb[0] + lb[1]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ls), __float_as_uint(ls), false, false);
    l_fin = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (npiece > 1) {
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
      p.ws_ml[prow * 2] = m_run;
      p.ws_ml[prow * 2 + 1] = l_fin;
    }
    return;
  }
  const float inv = 1.0f / l_fin;
  if (qrow < p.Lq) {
    T* op = (T*)p.out + (int64_t)b * p.o_sb + (int64_t)qrow * p.o_sl + (int64_t)h * p.o_sh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      f32x4 x0, x1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * g4 + i;
        x0[i] = (FOLD ? ot_lds[r * NT] : o0[r]) * inv;
        x1[i] = (FOLD ? ot_lds[(16 + r) * NT] : o1[r]) * inv;
      }
      *(v4*)(op + 8 * g4 + 4 * hi) = __builtin_convertvector(x0, v4);
      *(v4*)(op + 32 + 8 * g4 + 4 * hi) = __builtin_convertvector(x1, v4);
    }
    if (p.lse != nullptr && hi == 0)
      p.lse[((int64_t)b * p.H + h) * p.Lq + qrow] = m_run * p.scale + __logf(l_fin);
  }
}

template <typename T, bool FOLD>
hipError_t launch(const AttnKParams& p0, hipStream_t s) {
  AttnKParams p = p0;
  constexpr int QB = 256;
  p.nqb = (p.Lq + QB - 1) / QB;
  p.sk_items = p.B * p.H * p.nqb;
  p.sk_ix = (p.sk_items + 7) / 8;
  const int slots_x = 32;  // one 8-wave workgroup per CU
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
  hipLaunchKernelGGL((shared_attn_fwd_pp_kernel<T, FOLD>), dim3(grid), dim3(512), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || k <= 1) return e;
  return ir_launch_shared_attn_combine(p, std::is_same<T, __bf16>::value ? 1 : 0, QB, rem, s);
}

}  // namespace

hipError_t ir_launch_shared_attn_fwd_pp(const AttnKParams& p, int dtype, hipStream_t s) {
  const bool fold = (p.aa != nullptr);
  if (dtype == 1) return fold ? launch<__bf16, true>(p, s) : launch<__bf16, false>(p, s);
  return fold ? launch<_Float16, true>(p, s) : launch<_Float16, false>(p, s);
}

#endif  // IR_ABLATIONS
