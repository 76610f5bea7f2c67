// attn_probs.hip - materialise attention_probs (B,H,Lq,Lkv) for the dump path (gfx950), and the per-segment
// attention mass for consumers that only rank the references.
//
// face_replace/models/attn_processors.py:258-261 keeps the probability matrix when save_self_attentions is set
// (read by test.py:108, gradio_demo.py:118, coach.py:312).  The fused forward never forms it; these kernels recompute
// P = exp(scale*QK^T - LSE) from the LSE the fused forward emitted.
//
// attn_probs_lines_kernel (round 5; every shape whose segment lengths are multiples of 8 keys) - bound: HBM WRITES,
// H*L*Lkv*2 bytes per identity (0.84 GB at the 64x64-token layer, N = 4, t = 1).  The only thing that matters
// is the shape of the stores:
//   * the contraction is issued SWAPPED (S^T = K Q^T: keys on the MFMA rows, queries on its columns), and the key a lane
//     supplies for MFMA row i is keymap(i): after it a lane holds 16 CONSECUTIVE keys of ONE query row per 32 x 32
//     block (the un-swapped form of round 1 left a lane one key of 16 different rows: 2-byte stores, 64 B per row and
//     instruction - 12.5x the time per byte of 16-byte stores, MI355X_MICROARCH.md "stores of each flavour");
//   * the exponentiated block is rounded, packed and written to a wave-private LDS tile (rows x 128 B, 16-byte chunks
//     XOR-swizzled by the row: both directions conflict-free) and read back row-major, so that one store instruction
//     writes 8 whole 128-B lines (8 lanes x 16 B per row);
//   * a wave owns 32*NQ query rows: every K fragment loaded from L2 feeds NQ blocks, K is re-read L/(32*NQ) times from
//     L2 (= the bytes written, at NQ = 2), never from HBM: work items that share a K chunk are neighbours on one XCD;
//   * work items are (b, h, key chunk, row block): the key axis is cut so that the grid is >= ~16 rounds of the chip.
// No barriers (the LDS tile is private to the wave; LDS operations of one wave execute in order).
//
// attn_probs_generic_kernel (round 1) stays for segment lengths that are not multiples of 8 (the rows of P are then not
// 16-byte aligned): 2-byte stores, correct for everything.
//
// attn_segment_mass (round 5, opt-in): mass[b,h,i,s] = sum of the probabilities of row i over segment s
// ([self?] ++ ref 0 ++ ...), fp32 - what gradio_demo.py:119-127 reduces the 6.7 GB tensor to.  Same recompute, no big
// tensor: bound by the exponentials (B*H*L*Lkv of them), not by memory.
#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int PW = 4;  // waves per workgroup

struct ProbsPlan {
  int kc;         // keys per chunk (multiple of the kernel's step)
  int nch_self;   // chunks of the self segment (0 without it)
  int nch_ref;    // chunks per reference segment
  int nch_total;  // nch_self + N * nch_ref
  int nqb;        // row blocks of PW*32*NQ query rows per (b, h)
  int items;      // B * H * nch_total * nqb
};

// MFMA row i of a swapped 32 x 32 block carries key keymap(i) of the block: lane (lq, hi) then holds C rows
// (r&3) + 8(r>>2) + 4hi, r = 0..15  ==  keys 16*hi + r.
static __device__ __forceinline__ int keymap(int i) { return (i & 3) + 4 * (i >> 3) + 16 * ((i >> 2) & 1); }

template <typename T>
static __device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  T2 v;
  v[0] = (T)a;
  v[1] = (T)b;
  return __builtin_bit_cast(unsigned, v);
}

template <typename T, int NQ, int NK, bool MASS>
__global__ void __launch_bounds__(PW * 64) attn_probs_lines_kernel(const AttnKParams p, const ProbsPlan pl, float* __restrict__ mass) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  constexpr int ROWS = 32 * NQ;                   // query rows per wave
  constexpr int STEP = 32 * NK;                   // keys per step
  constexpr int PITCH = 2 * STEP;                 // bytes of one row of the wave's LDS tile (NK * 64)
  constexpr int TILE = MASS ? 16 : ROWS * PITCH;  // LDS bytes per wave
  constexpr int LPR = PITCH / 16;                 // lanes that read one row back (16 B each)
  constexpr int RPI = 64 / LPR;                   // rows per store instruction
  constexpr int NST = ROWS / RPI;                 // store instructions per step
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[PW * TILE];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, lq = lane & 31;

  // work item: row block fastest (neighbours on one XCD read the same K chunk), then key chunk, then (b, h)
  const int id = xcd_remap(blockIdx.x, pl.items);
  const int qb = id % pl.nqb;
  const int t0 = id / pl.nqb;
  const int c = t0 % pl.nch_total;
  const int bh = t0 / pl.nch_total;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = (qb * PW + wid) * ROWS;
  if (q0 >= p.Lq) return;

  const T* kb;
  int64_t ksl;
  int len, col0, chunk, seg;
  if (c < pl.nch_self) {
    kb = (const T*)p.k_self + (int64_t)b * p.ks_sb + (int64_t)h * p.ks_sh; ksl = p.ks_sl; len = p.Ls; col0 = 0; chunk = c; seg = 0;
  } else {
    const int cr = c - pl.nch_self;
    const int n = cr / pl.nch_ref;
    chunk = cr - n * pl.nch_ref;
    kb = (const T*)p.k_ref + (int64_t)b * p.kr_sb + (int64_t)n * p.kr_sn + (int64_t)h * p.kr_sh; ksl = p.kr_sl; len = p.Lr;
    col0 = p.include_self * p.Ls + n * p.Lr;
    seg = p.include_self + n;
  }
  const int j_begin = chunk * pl.kc;
  const int j_end = (j_begin + pl.kc < len) ? j_begin + pl.kc : len;

  // B operand: Q[row 32*qi + lq][d = 16ks + 8hi ..]; the LSE of that row in the exp2 domain
  const float LOG2E = 1.4426950408889634f;
  v8 qf[NQ][4];
  float lse2[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = q0 + 32 * qi + lq;
    const int qr = q < p.Lq ? q : p.Lq - 1;
    const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)qr * p.q_sl + (int64_t)h * p.q_sh + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qi][ks] = *(const v8*)(qp + ks * 16);
    lse2[qi] = p.lse[((int64_t)b * p.H + h) * p.Lq + qr] * LOG2E;
  }

  // A operand: K[key j + 32*kblk + keymap(lq)][d = 16ks + 8hi ..], keys past the segment clamped (their columns are not stored)
  const int km = keymap(lq);
  v8 kf[NK][4];
  auto load_k = [&](int j, int kblk) {   // one 32-key block of fragments
    const int key = j + 32 * kblk + km;
    const T* kp = kb + (int64_t)(key < len ? key : len - 1) * ksl + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[kblk][ks] = *(const v8*)(kp + ks * 16);
  };
#pragma unroll
  for (int kblk = 0; kblk < NK; ++kblk) load_k(j_begin, kblk);

  unsigned char* tile = lds_all + wid * TILE;
  // read-back role of this lane: row RPI*t + lane / LPR, 16-byte chunk lane % LPR (8 keys) of the step.  Stores go through ONE
  // buffer resource over the workgroup's rows of P (<= PW*ROWS rows: offsets stay below 2^31, launch-side check): rows past
  // Lq fall outside num_records and are dropped by the range check, key chunks past the segment get an offset that is - so
  // the store loop has no branch, and hipcc counts the stores in flight instead of draining them around a branch
  const int rrow = lane / LPR, rchunk = lane % LPR;
  const int wg_row0 = qb * (PW * ROWS);
  const int wg_rows = (p.Lq - wg_row0 < PW * ROWS) ? p.Lq - wg_row0 : PW * ROWS;
  __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((T*)p.probs + (((int64_t)b * p.H + h) * p.Lq + wg_row0) * (int64_t)p.lkv), 0, (unsigned)wg_rows * (unsigned)p.lkv * 2u, 0x00020000);
  const unsigned row_pitch = (unsigned)p.lkv * 2u * RPI;   // RPI rows of P in bytes
  unsigned ooff = ((unsigned)(wid * ROWS + rrow) * (unsigned)p.lkv + (unsigned)(col0 + j_begin + rchunk * 8)) * 2u;
  float msum[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) msum[qi] = 0.f;

  for (int j = j_begin; j < j_end; j += STEP) {
#pragma unroll
    for (int kblk = 0; kblk < NK; ++kblk) {
      f32x16 sc[NQ];
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[qi][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) sc[qi] = Tr::mfma(kf[kblk][ks], qf[qi][ks], sc[qi]);
      }
      // the block's fragments are dead: the next step's arrive in the same registers while this step's exponentials run
      // (clamped past the end of the segment: always a valid address)
      load_k(j + STEP, kblk);
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        // sc[qi][r] = <K[j + 32 kblk + 16 hi + r], Q[q0 + 32 qi + lq]>
        if constexpr (MASS) {
          const int kfirst = j + 32 * kblk + 16 * hi;
          if (j + STEP <= j_end) {   // wave-uniform: whole steps need no key mask
#pragma unroll
            for (int r = 0; r < 16; ++r) msum[qi] += fast_exp2(__builtin_fmaf(sc[qi][r], p.scale_log2, -lse2[qi]));
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float e = fast_exp2(__builtin_fmaf(sc[qi][r], p.scale_log2, -lse2[qi]));
              msum[qi] += (kfirst + r < j_end) ? e : 0.f;
            }
          }
        } else {
          unsigned w[8];
#pragma unroll
          for (int r = 0; r < 16; r += 2)
            w[r >> 1] = pack2<T>(fast_exp2(__builtin_fmaf(sc[qi][r], p.scale_log2, -lse2[qi])),
                                 fast_exp2(__builtin_fmaf(sc[qi][r + 1], p.scale_log2, -lse2[qi])));
          const int R = 32 * qi + lq;
          unsigned char* rowp = tile + R * PITCH;
          const int c0 = kblk * 4 + hi * 2;
          *(u32x4_alias*)(rowp + (((c0 + 0) ^ (R & 7)) << 4)) = u32x4{w[0], w[1], w[2], w[3]};
          *(u32x4_alias*)(rowp + (((c0 + 1) ^ (R & 7)) << 4)) = u32x4{w[4], w[5], w[6], w[7]};
        }
      }
    }
    if constexpr (!MASS) {
      ir_wave_lds_fence();
      const unsigned obase = (j + rchunk * 8 < j_end) ? ooff : 0x80000000u;
      u32x4 v[NST];
#pragma unroll
      for (int t = 0; t < NST; ++t) {
        const int R = RPI * t + rrow;
        v[t] = *(const u32x4_alias*)(tile + R * PITCH + ((rchunk ^ (R & 7)) << 4));
      }
      ir_wave_lds_fence();   // every read issued before the first store waits for its data
#pragma unroll
      // nt (non-temporal, aux bit 1 on gfx940+): P is written once and never read here, K is re-read by every wave from L2 - the
      // streaming hint keeps the 6.7 GB of P from pushing K out of the 4-MiB L2s.  A/B of the cache-policy bits on one box, top
      // layer (round-5 A/B driver, profiles/r5_probs_probe.txt): plain 1.69-1.74 ms, nt 1.49-1.53, sc1 / sc0 sc1 (write-through) 1.70-1.74, sc1 nt
      // 1.53; whole probe on the next box (profiles/r5_probs_probe.txt): 1.23 ms = 5.4 TB/s, L = 1024 0.26 -> 0.18 ms.
      // -DPROBS_STORE_AUX=<bits> rebuilds the A/B (1 sc0, 2 nt, 16 sc1).
#ifndef PROBS_STORE_AUX
#define PROBS_STORE_AUX 2
#endif
      for (int t = 0; t < NST; ++t) __builtin_amdgcn_raw_buffer_store_b128(v[t], prs, obase + (unsigned)t * row_pitch, 0, PROBS_STORE_AUX);
      ooff += (unsigned)PITCH;
    }
  }

  if constexpr (MASS) {
    const int nseg = p.include_self + p.N;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      const float s = msum[qi] + __shfl_xor(msum[qi], 32);
      const int q = q0 + 32 * qi + lq;
      if (hi == 0 && q < p.Lq) mass[(((int64_t)b * p.H + h) * p.Lq + q) * nseg + seg] = s;
    }
  }
}

// ---- round 1's kernel: any segment length (2-byte stores) -------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(PW * 64) attn_probs_generic_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, lq = lane & 31;
  const int nqb = (p.Lq + PW * 32 - 1) / (PW * 32);
  const int bh = blockIdx.x / nqb;
  const int qb = blockIdx.x - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * (PW * 32) + wid * 32;
  if (q0 >= p.Lq) return;

  // A operand: Q[row lq][d = 16ks + 8hi ..]
  const int qr = (q0 + lq < p.Lq) ? q0 + lq : p.Lq - 1;
  const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)qr * p.q_sl + (int64_t)h * p.q_sh + hi * 8;
  v8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8*)(qp + ks * 16);

  // output rows of this lane: q0 + crow(r,hi); their LSE in the exp2 domain
  const float LOG2E = 1.4426950408889634f;
  float lse2[16];
  int64_t orow[16];
  bool rok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    rok[r] = q < p.Lq;
    const int qc = rok[r] ? q : p.Lq - 1;
    lse2[r] = p.lse[((int64_t)b * p.H + h) * p.Lq + qc] * LOG2E;
    orow[r] = (((int64_t)b * p.H + h) * p.Lq + qc) * (int64_t)p.lkv;
  }
  T* probs = (T*)p.probs;

  const int nseg = p.include_self + p.N;
  int col0 = 0;
  for (int s = 0; s < nseg; ++s) {
    const T* kb;
    int64_t ksl;
    int len;
    if (p.include_self && s == 0) {
      kb = (const T*)p.k_self + (int64_t)b * p.ks_sb + (int64_t)h * p.ks_sh; ksl = p.ks_sl; len = p.Ls;
    } else {
      const int n = s - p.include_self;
      kb = (const T*)p.k_ref + (int64_t)b * p.kr_sb + (int64_t)n * p.kr_sn + (int64_t)h * p.kr_sh; ksl = p.kr_sl; len = p.Lr;
    }
    for (int j0 = 0; j0 < len; j0 += 32) {
      const int key = j0 + lq;
      const bool kok = key < len;
      const T* kp = kb + (int64_t)(kok ? key : len - 1) * ksl + hi * 8;
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc = Tr::mfma(qf[ks], *(const v8*)(kp + ks * 16), sc);
      // sc[r] = <Q[q0+crow(r,hi)], K[key]> ; column = lane & 31 = key
      if (kok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (rok[r]) probs[orow[r] + col0 + key] = (T)fast_exp2(__builtin_fmaf(sc[r], p.scale_log2, -lse2[r]));
        }
      }
    }
    col0 += len;
  }
}

static int device_cus() {   // per device, looked up once (idempotent: a race between threads only repeats the query)
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    cached[dev] = cus;
  }
  return cached[dev];
}

// Cut of the key axis: the largest chunk (whole segments first) that still leaves >= 16 workgroups per CU; never below 256 keys.
static ProbsPlan make_plan(const AttnKParams& p, int rows_per_wg, int step, bool whole_segments) {
  ProbsPlan pl;
  pl.nqb = (p.Lq + rows_per_wg - 1) / rows_per_wg;
  const int maxlen = (p.include_self && p.Ls > p.Lr) || p.N == 0 ? p.Ls : p.Lr;
  int kc = (maxlen + step - 1) / step * step;
  if (!whole_segments) {
    const int64_t want = (int64_t)16 * device_cus();
    for (;;) {
      const int64_t ns = p.include_self ? (p.Ls + kc - 1) / kc : 0;
      const int64_t nr = p.N > 0 ? (p.Lr + kc - 1) / kc : 0;
      const int64_t items = (int64_t)p.B * p.H * pl.nqb * (ns + (int64_t)p.N * nr);
      if (items >= want || kc <= 256) break;
      kc = ((kc / 2) + step - 1) / step * step;
    }
  }
  pl.kc = kc;
  pl.nch_self = p.include_self ? (p.Ls + kc - 1) / kc : 0;
  pl.nch_ref = p.N > 0 ? (p.Lr + kc - 1) / kc : 0;
  pl.nch_total = pl.nch_self + p.N * pl.nch_ref;
  const int64_t items = (int64_t)p.B * p.H * pl.nqb * pl.nch_total;
  pl.items = items > 0x7fffffffLL ? -1 : (int)items;
  return pl;
}

static bool lines_ok(const AttnKParams& p) {
  if (p.lkv % 8) return false;
  if (p.include_self && p.Ls % 8) return false;
  if (p.N > 0 && p.Lr % 8) return false;
  if ((int64_t)p.lkv * 2 * (PW * 64 + 8) >= (int64_t)1 << 31) return false;   // 32-bit store offsets within a workgroup's rows
  return ((uintptr_t)p.probs & 15) == 0;
}

}  // namespace

bool ir_attn_probs_uses_lines(const AttnKParams& p) { return lines_ok(p); }

namespace {
template <typename T, int NQ, int NK>
hipError_t launch_lines(const AttnKParams& p, hipStream_t s) {
  const ProbsPlan pl = make_plan(p, PW * 32 * NQ, 32 * NK, false);
  if (pl.items <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL((attn_probs_lines_kernel<T, NQ, NK, false>), dim3(pl.items), dim3(PW * 64), 0, s, p, pl, (float*)nullptr);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_lines_variant(const AttnKParams& p, int variant, hipStream_t s) {
  switch (variant) {
    case 2: return launch_lines<T, 2, 2>(p, s);   // 64 rows x 128 B per wave and step
    case 3: return launch_lines<T, 1, 2>(p, s);   // 32 rows x 128 B
    case 4: return launch_lines<T, 1, 4>(p, s);   // 32 rows x 256 B
    case 5: return launch_lines<T, 2, 4>(p, s);   // 64 rows x 256 B
    case 6: return launch_lines<T, 1, 8>(p, s);   // 32 rows x 512 B
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

hipError_t ir_launch_attn_probs(const AttnKParams& p, int dtype, int variant, hipStream_t s) {
  if (variant != 1 && lines_ok(p)) {
    // automatic: 64 rows per wave; 32 when the rows of one (b, h) would not fill a 256-row workgroup (the 16 x 16-token class)
    const int v = variant >= 2 ? variant : (p.Lq >= 256 ? 2 : 3);
    return dtype == 1 ? launch_lines_variant<__bf16>(p, v, s) : launch_lines_variant<_Float16>(p, v, s);
  }
  if (variant >= 2) return hipErrorInvalidValue;   // the line kernel was asked for by name and does not cover the shape
  const int nqb = (p.Lq + PW * 32 - 1) / (PW * 32);
  const dim3 grid(p.B * p.H * nqb);
  if (dtype == 1) hipLaunchKernelGGL((attn_probs_generic_kernel<__bf16>), grid, dim3(PW * 64), 0, s, p);
  else hipLaunchKernelGGL((attn_probs_generic_kernel<_Float16>), grid, dim3(PW * 64), 0, s, p);
  return hipGetLastError();
}

hipError_t ir_launch_attn_segment_mass(const AttnKParams& p, int dtype, float* mass, hipStream_t s) {
  const bool nq2 = p.Lq >= 256;
  const ProbsPlan pl = make_plan(p, PW * (nq2 ? 64 : 32), 64, true);   // whole segments: one writer per (row, segment), no reduction through memory
  if (pl.items <= 0) return hipErrorInvalidValue;
  const dim3 grid(pl.items), block(PW * 64);
  if (dtype == 1) {
    if (nq2) hipLaunchKernelGGL((attn_probs_lines_kernel<__bf16, 2, 2, true>), grid, block, 0, s, p, pl, mass);
    else hipLaunchKernelGGL((attn_probs_lines_kernel<__bf16, 1, 2, true>), grid, block, 0, s, p, pl, mass);
  } else {
    if (nq2) hipLaunchKernelGGL((attn_probs_lines_kernel<_Float16, 2, 2, true>), grid, block, 0, s, p, pl, mass);
    else hipLaunchKernelGGL((attn_probs_lines_kernel<_Float16, 1, 2, true>), grid, block, 0, s, p, pl, mass);
  }
  return hipGetLastError();
}
