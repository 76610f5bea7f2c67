// linear_tiled.hip - Y[M,N] = X[M,K] W[N,K]^T (+ bias) for every projection shape the X-stationary kernels of
// linear_skinny.hip do not cover: K = 1280 (the 16x16-token layer class: to_q/k/v fused N = 3840, to_out N = 1280), and
// the small-M shapes of every class (8 identities: M = 2048 / 8192 / 32768 rows), gfx950.  Round 3: these were the
// last vendor GEMMs of the step (attn_processors.py:222-230,267).
//
// Shape of the kernel: a workgroup owns a BM x BN tile of Y and walks K in 64-wide steps; both operand tiles live in LDS
// as 128-byte rows whose 16-byte slots are XOR-swizzled with (row >> 1) & 7 (the attention kernels' K-tile image:
// conflict-free ds_read_b128 of the MFMA fragments), two stages.  16-bit operands arrive by LDS-DMA (buffer_load ... lds
// issued from asm, the swizzle applied to the SOURCE slot a lane fetches; ragged row tails are zero-filled by the
// descriptor's bounds check).  fp32 activations (the LayerNorm output under autocast, inference/test.py:83) cannot ride
// the DMA - it moves bytes - so they are loaded to registers one K-step ahead, rounded (RNE: the bytes `.to(dtype)` would
// produce) after the step's MFMAs and written with ds_write_b128: the cast costs no pass over memory.
// Products are issued "swapped" (Y^T = W X^T) on v_mfma_f32_32x32x16 like everywhere in this library: a lane owns one
// row of Y and 16 of its columns per 32x32 block, so the epilogue (column scale of the pre-scaled-Q contract, bias,
// ONE rounding) stages 64 x 64 per wave in LDS and leaves as whole 128-byte lines of Y.
// Tile -> workgroup map: every XCD owns a contiguous range of tiles (its L2 sees each operand panel once), and inside
// it tiles are ordered 8 row panels x all column tiles, so the ~64 tiles resident on an XCD at a time form an 8 x 8
// patch that shares its X and W panels in that L2.
// Deterministic: no atomics, no split-K; the accumulation order of an output element is fixed by the tile shape.
#include <stdlib.h>
#include <type_traits>

#include "ir_common.h"
#include "ir_colstats.h"
#include "ir_kernels.h"

namespace {

constexpr int kTiledPitch = 144;   // staging tile row: 128 B (64 columns of Y) + 16 B pad

// KW = 2 (round 5): TWO wave groups per workgroup split the contraction - group g multiplies K-steps [g KT/2, (g+1) KT/2) of
// the SAME tile out of its own pair of LDS stages, the groups meet at the end: group 1 hands its fp32 accumulators over
// through LDS, group 0 adds them (always k-half 0 + k-half 1: one fixed order, no atomics, nothing through memory) and
// runs the epilogue.  For shapes whose tile count leaves CUs idle (M = 2048 rows: 160 tiles of 128 x 128 at N = 1280): the K
// loop of a tile is half as long and every busy CU holds eight waves instead of four.
template <typename T, int WM, int WN, int MI, int NI, bool XF32, bool ILV, int KW = 1>
__global__ void __launch_bounds__(WM * WN * KW * 64, KW == 1 ? 2 : 1) linear_tiled_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NW = WM * WN, NT = NW * 64;      // waves / threads of ONE K group
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr int XT = BM * 128, WT = BN * 128, STAGE = XT + WT;
  constexpr int XP = BM / 8 / NW, WP = BN / 8 / NW;    // 1-KiB LDS-DMA pieces per wave and K-step
  constexpr int XU = BM * 8 / NT;                      // fp32 path: 8-element units per thread and K-step
  static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "pieces must divide over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_tiled[];
  const int lane = threadIdx.x & 63;
  const int wid_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kg = KW == 1 ? 0 : wid_all / NW;     // K group of this wave
  const int wid = KW == 1 ? wid_all : wid_all - kg * NW;
  const int tid = wid * 64 + lane;               // thread index inside the group
  unsigned char* const smem = dsm_tiled + kg * (2 * STAGE);   // the group's own two stages

  const int wm = wid / WN, wn = wid - wm * WN;
  const int hi = lane >> 5, lq = lane & 31;

  // ---- tile decode -----------------------------------------------------------------------------------------------
  const int MT = (p.M + BM - 1) / BM, NTl = p.N / BN;
  const int logical = xcd_remap((int)blockIdx.x, MT * NTl);
  const int GM = p.nsplit;        // row panels per walk group (launch_cfg: 8; IR_LIN_GM for the PMC A/B of profiles/r4_pmc_linear_tiled.txt)
  const int grp = logical / (GM * NTl), rem = logical - grp * (GM * NTl);
  const int gm = (MT - grp * GM) < GM ? (MT - grp * GM) : GM;
  const int tm = grp * GM + rem % gm, tn = rem / gm;
  const int m0 = tm * BM, n0 = tn * BN;
  const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
  const int KT = (p.K >> 6) / KW;                // K-steps of this group (the launcher picks KW = 2 only for an even count)
  const int kt0 = kg * KT;                       // ... starting here

  // ---- operand streams ---------------------------------------------------------------------------------------------
  const T* const wbase = (const T*)p.w + (int64_t)n0 * p.w_ld;
  const i32x4 wrs = make_rsrc_words(wbase, (unsigned)(((int64_t)(BN - 1) * p.w_ld + p.K) * 2));
  unsigned wvo[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int row = 8 * (wid + j * NW) + (lane >> 3);
    wvo[j] = (unsigned)(row * p.w_ld * 2 + ((((lane & 7) ^ ((row >> 1) & 7))) << 4));
  }
  i32x4 xrs = wrs;
  unsigned xvo[XF32 ? 1 : XP];
  const float* xfp[XF32 ? XU : 1];
  int xlds[XF32 ? XU : 1];
  if constexpr (!XF32) {
    const T* const xbase = (const T*)p.x + (int64_t)m0 * p.x_ld;
    xrs = make_rsrc_words(xbase, (unsigned)(((int64_t)(rows_valid - 1) * p.x_ld + p.K) * 2));
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int row = 8 * (wid + j * NW) + (lane >> 3);
      xvo[j] = (unsigned)(row * p.x_ld * 2 + ((((lane & 7) ^ ((row >> 1) & 7))) << 4));
    }
  } else {
#pragma unroll
    for (int j = 0; j < XU; ++j) {
      const int u = tid + j * NT, row = u >> 3, ch = u & 7;
      const int gr = (m0 + row) < p.M ? (m0 + row) : (p.M - 1);   // rows past M: copies of the last row (never stored)
      xfp[j] = (const float*)p.x + (int64_t)gr * p.x_ld + ch * 8;
      xlds[j] = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
    }
  }
  // one 1-KiB piece of the next stage: pieces [0, WP) are W's, [WP, WP + XP) X's (16-bit activations only)
  constexpr int NPIECE = WP + (XF32 ? 0 : XP);
  auto issue_piece = [&](int j, int kt, int slot) {
    unsigned char* const sx = smem + slot * STAGE;
    if (j < WP) buffer_load_lds16_async(wrs, sx + XT + (wid + j * NW) * 1024, wvo[j] + (kt0 + kt) * 128);
    else if constexpr (!XF32) buffer_load_lds16_async(xrs, sx + (wid + (j - WP) * NW) * 1024, xvo[j - WP] + (kt0 + kt) * 128);
  };
  auto issue_dma = [&](int kt, int slot) {
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) issue_piece(j, kt, slot);
  };
  // fp32 activations, one K-step ahead in registers.  The loads are issued from asm: hipcc sinks a plain load to its
  // first use - BEHIND the step's MFMAs, where it is waited for at once (measured: 1.5x the 16-bit path) - and waits
  // for a volatile one immediately.  An asm load is invisible to hipcc's waitcnt bookkeeping, so the statement that
  // waits for them names every destination register as read-write: no consumer can be scheduled above it.
  f32x4 xr[XF32 ? XU : 1][2];
  auto load_x32 = [&](int kt) {
    if constexpr (XF32) {
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xr[j][0]) : "v"(xfp[j] + (kt0 + kt) * 64) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(xr[j][1]) : "v"(xfp[j] + (kt0 + kt) * 64) : "memory");
      }
    }
  };
  auto put_x32 = [&](int slot) {
    if constexpr (XF32) {
      unsigned char* const sx = smem + slot * STAGE;
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0][0]), "+v"(xr[0][1]) : : "memory");
#pragma unroll
      for (int j = 1; j < XU; ++j) asm volatile("" : "+v"(xr[j][0]), "+v"(xr[j][1]));
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const f32x8 f = __builtin_shufflevector(xr[j][0], xr[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        *(IR_LDS v8*)(IR_LDS unsigned char*)(sx + xlds[j]) = __builtin_convertvector(f, v8);
      }
    }
  };

  // ---- fragment addresses ------------------------------------------------------------------------------------------
  int fread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fread[ks] = lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);
  const int xrow0 = wm * (MI * 32) * 128, wrow0 = XT + wn * (NI * 32) * 128;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // The MFMAs of one K-step.  `dma_kt >= 0`: the LDS-DMA pieces of stage dma_kt are issued BETWEEN the MFMAs of the
  // first two k-substeps, one piece per `STRIDE` MFMAs.  Issued in a bunch at the top of the step they cost each wave
  // ~100 issue cycles apiece with the matrix pipe idle - both waves of a SIMD leave the barrier in the same phase -
  // (measured: the 256x256 tile ran at ~50 % matrix-pipe occupancy); between MFMAs they ride in the shadow of the
  // 32-cycle matrix instructions, and issuing them in the first half leaves the second half for them to land.
  constexpr int HALF_MFMA = 2 * MI * NI;
  constexpr int STRIDE = (HALF_MFMA / NPIECE) > 0 ? (HALF_MFMA / NPIECE) : 1;
  auto compute = [&](int slot, int dma_kt, int dma_slot) {
    const unsigned char* const st = smem + slot * STAGE;
    v8 wf[2][NI], xf[2][MI];
    auto rd = [&](int ks, int b) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[b][ni] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(st + wrow0 + ni * 4096 + fread[ks]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[b][mi] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(st + xrow0 + mi * 4096 + fread[ks]);
    };
    rd(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) rd(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = Tr::mfma(wf[ks & 1][ni], xf[ks & 1][mi], acc[mi][ni]);
          if constexpr (ILV) {
            const int m = ks * MI * NI + mi * NI + ni;           // MFMA index within the step (compile-time after unrolling)
            if (m < HALF_MFMA && (m % STRIDE) == STRIDE - 1 && (m / STRIDE) < NPIECE) {
              __builtin_amdgcn_sched_barrier(0);
              if (dma_kt >= 0) issue_piece(m / STRIDE, dma_kt, dma_slot);
              __builtin_amdgcn_sched_barrier(0);
            }
            if (m == HALF_MFMA - 1) {                            // pieces that did not fit a slot of their own
#pragma unroll
              for (int j = HALF_MFMA / STRIDE; j < NPIECE; ++j)
                if (dma_kt >= 0) issue_piece(j, dma_kt, dma_slot);
            }
          }
        }
    }
  };

  // ---- main loop: stage kt+1 while kt is multiplied --------------------------------------------------------------
  issue_dma(0, 0);
  load_x32(0);
  put_x32(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt + 1 < KT; ++kt) {
    const int cur = kt & 1;
    if constexpr (!ILV) issue_dma(kt + 1, cur ^ 1);   // that stage was last read in step kt-1, behind the barrier that ended it
    load_x32(kt + 1);
    compute(cur, kt + 1, cur ^ 1);
    put_x32(cur ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  compute((KT - 1) & 1, -1, 0);
  __syncthreads();                  // every wave is done with the operand stages: they become the output staging tiles
  if constexpr (KW == 2) {
    // hand-over of the second K half: wave w of group 1 leaves its MI*NI*16 accumulators, four per lane and slot, in the upper
    // half of LDS (behind everything the epilogue's staging tiles touch); wave w of group 0 adds them in a fixed order
    constexpr int HAND_OFF = 2 * STAGE;                  // = group 1's own stages: free since the barrier above
    static_assert(2 * STAGE >= NW * MI * 32 * kTiledPitch, "epilogue staging must stay below the hand-over area");
    static_assert(2 * STAGE >= NW * MI * NI * 16 * 64 * 4, "hand-over area");
    unsigned char* const hand = dsm_tiled + HAND_OFF + wid * (MI * NI * 16 * 64 * 4) + lane * 16;
    if (kg == 1) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(f32x4_alias*)(hand + ((mi * NI + ni) * 4 + g) * 1024) = f32x4{acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 o = *(const f32x4_alias*)(hand + ((mi * NI + ni) * 4 + g) * 1024);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mi][ni][4 * g + i] += o[i];
        }
  }

  // ---- epilogue: column scale, bias, one rounding, transposed through LDS into whole lines of Y -------------------
  // (64 columns at a time: a wave's LDS operations execute in order, so the tile is reused without a barrier)
  unsigned char* const tb = smem + wid * (MI * 32 * kTiledPitch);
  const int row_base = m0 + wm * (MI * 32);
  v4 bvs[NI][4];   // the lane's 16 bias values per 32-column block, fetched together ahead of the loops
  if (p.bias != nullptr) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) bvs[ni][g] = *(const v4*)((const T*)p.bias + n0 + wn * (NI * 32) + ni * 32 + 8 * g + 4 * hi);
  } else {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int g = 0; g < 4; ++g) bvs[ni][g] = v4{0, 0, 0, 0};
  }
#pragma unroll
  for (int nh = 0; nh < NI / 2; ++nh) {
    const int ncol0 = n0 + wn * (NI * 32) + nh * 64;
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2) {
      const int ni = 2 * nh + n2;
      const float cs = (ncol0 + n2 * 32) < p.scale_cols ? p.col_scale : 1.0f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 bz;
#pragma unroll
        for (int i = 0; i < 4; ++i) bz[i] = (float)bvs[ni][g][i];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          f32x4 f;
#pragma unroll
          for (int i = 0; i < 4; ++i) f[i] = acc[mi][ni][4 * g + i] * cs + bz[i];
          *(v4*)(tb + (mi * 32 + lq) * kTiledPitch + n2 * 64 + (8 * g + 4 * hi) * 2) = __builtin_convertvector(f, v4);
        }
      }
    }
    ir_wave_lds_fence();   // the reads below fetch what OTHER lanes have just written
    // round 4: the finished 16-bit rows pass through the lanes here, eight columns of row 8 j + (lane >> 3) each - the shape
    // the token statistics of a V head are accumulated in (ir_colstats.h): no load, no pass over V
    const int st_head = (ncol0 - p.st_col0) >> 6;
    const bool st_on = p.st_ws != nullptr && ncol0 >= p.st_col0 && st_head < (p.st_cols >> 6) && row_base < p.M;   // wave-uniform
    ColStatsAcc sacc;
#pragma unroll
    for (int j = 0; j < MI * 4; ++j) {
      const int r = 8 * j + (lane >> 3);
      const u32x4 v = *(const u32x4_alias*)(tb + r * kTiledPitch + (lane & 7) * 16);
      const int row = row_base + r;
      if (row < p.M) ir_store_y((u32x4*)((T*)p.y + (int64_t)row * p.y_ld + ncol0 + (lane & 7) * 8), v);
      if (st_on) {
        if (j == 0) ir_stats_first<T>(sacc, v);
        else ir_stats_add<T>(sacc, v);
      }
    }
    ir_wave_lds_fence();   // ... and the next 64-column group's writes land on what other lanes have just read
    if (st_on) ir_stats_finish(sacc, p.st_ws + ((int64_t)(row_base / kStatsRows) * (p.st_cols >> 6) + st_head) * 128);
  }
}


// ---- 256 x 256 tile, PING-PONG wave groups ------------------------------------------------------------------------
// Same tile, same LDS images and operand streams as linear_tiled_kernel<T, 4, 2, 2, 4, ...>; what changes is WHEN a wave
// does what.  There, all eight waves leave the step's barrier in the same phase: both waves of a SIMD read fragments
// together (matrix pipe idle) and then issue MFMAs together (each waits for the other's) - matrix time and
// load time add up (profiles/r3_gemm_ablation.txt: 1.58 us per step for 1.02 us of MFMAs).  Here a K-step is four
// barrier-separated phases per wave,
//     M0 (8 + 8 MFMAs of k-substeps 0, 1) | L1 (fragments of k-substeps 2, 3) | M1 (their MFMAs) | L0' (fragments of the
//     NEXT step's k-substeps 0, 1 + its LDS-DMA issue)
// and waves 4-7 run ONE PHASE behind waves 0-3 (one extra barrier at their start, one at the others' end): a workgroup's
// waves land on the SIMDs cyclically, so every SIMD has one wave in a matrix phase (16 MFMAs back to back, the pipe to
// itself) beside one in a load phase (12 ds_read_b128, 8 LDS-DMA pieces or the fp32 loads / rounding / ds_write).
// LDS-DMA of step k+1 is issued in L0(k) - its stage was last read in L1(k-1) of both groups, two barriers earlier - and
// waited for (vmcnt(0), own pieces) at the end of L1(k); the first read of that stage is L0(k+1) of group A, one barrier
// after group B's wait.  Every load phase ends with lgkmcnt(0): its fragment reads have left LDS before the barrier
// after which another wave's DMA may overwrite that stage.
template <typename T, bool XF32>
__global__ void __launch_bounds__(512, 2) linear_tiled_pp_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NW = 8, NT = 512, WN = 2, MI = 2, NI = 4;
  constexpr int BM = 256, BN = 256;
  constexpr int XT = BM * 128, WT = BN * 128, STAGE = XT + WT;
  constexpr int XP = BM / 8 / NW, WP = BN / 8 / NW;    // 4 + 4 one-KiB pieces per wave and K-step
  constexpr int XU = BM * 8 / NT;                      // fp32 path: 4 eight-element units per thread and K-step
  constexpr int OUT_OFF = 2 * STAGE;                   // wave-private output staging (4 KiB each) BEHIND the stages
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_tiled_pp[];
  unsigned char* const smem = dsm_tiled_pp;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2;                            // 0: waves 0-3, 1: waves 4-7 (one phase behind)
  const int wm = wid / WN, wn = wid - wm * WN;
  const int hi = lane >> 5, lq = lane & 31;
  const int KT = p.K >> 6;

  // ---- tile walk: a workgroup takes tiles xs, xs + (workgroups on its XCD), ... of its XCD's contiguous range - exactly one
  // when the launch has a workgroup per tile (tile counts up to the CU count), several when it is persistent (one workgroup per
  // CU, the default for larger tile counts since round 4; then the next tile's first stage is in flight during the epilogue).
  // The ~6.5 us per tile beyond the K loop are the output write (HBM-bound burst), not launch or a cold first stage
  // (profiles/r3_gemm_ablation.txt); the persistent form hides a little of it: 1-3 % per multi-round GEMM (launch_pp)
  const int MT = (p.M + BM - 1) / BM, NTl = (p.N + BN - 1) / BN;   // the last column tile may be ragged (N % 64 == 0)
  const int ntiles = MT * NTl;
  const int xcd = blockIdx.x & 7, xs = blockIdx.x >> 3;
  const int q8 = ntiles >> 3, r8 = ntiles & 7;
  const int t_start = (xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;   // xcd_remap's ranges
  const int t_count = q8 + (xcd < r8 ? 1 : 0);
  const int wg_x = ((int)gridDim.x >> 3) + (xcd < ((int)gridDim.x & 7) ? 1 : 0);       // workgroups of this launch on this XCD

  // tile-independent lane coordinates
  unsigned wvo[WP], xvo[XF32 ? 1 : XP];
  int xlds0 = 0;                                       // fp32 path: unit j sits 64 rows (8 KiB, same swizzle) below unit 0
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    const int row = 8 * (wid + j * NW) + (lane >> 3);
    wvo[j] = (unsigned)(row * p.w_ld * 2 + ((((lane & 7) ^ ((row >> 1) & 7))) << 4));
  }
  if constexpr (!XF32) {
#pragma unroll
    for (int j = 0; j < XP; ++j) {
      const int row = 8 * (wid + j * NW) + (lane >> 3);
      xvo[j] = (unsigned)(row * p.x_ld * 2 + ((((lane & 7) ^ ((row >> 1) & 7))) << 4));
    }
  } else {
    const int row = tid >> 3, ch = tid & 7;
    xlds0 = row * 128 + ((ch ^ ((row >> 1) & 7)) << 4);
  }
  int fread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) fread[ks] = lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);
  const int xrow0 = wm * (MI * 32) * 128, wrow0 = XT + wn * (NI * 32) * 128;

  // ---- per-tile operand streams -------------------------------------------------------------------------------------------
  int m0 = 0, n0 = 0;
  i32x4 wrs = {0, 0, 0, 0}, xrs = {0, 0, 0, 0};
  const float* xtile = nullptr;                        // fp32 path: first row of the tile (wave-uniform) ...
  unsigned xoff[XF32 ? XU : 1];                        // ... and the thread's byte offsets from it (32 bit: a tile spans < 4 GiB)
  auto set_tile = [&](int logical) {
    const int GM = p.nsplit;      // row panels per walk group (launch_pp)
    const int tgrp = logical / (GM * NTl), rem = logical - tgrp * (GM * NTl);
    const int gm = (MT - tgrp * GM) < GM ? (MT - tgrp * GM) : GM;
    const int tm = tgrp * GM + rem % gm, tn = rem / gm;
    m0 = tm * BM; n0 = tn * BN;
    const int rows_valid = (p.M - m0) < BM ? (p.M - m0) : BM;
    const int cols_valid = (p.N - n0) < BN ? (p.N - n0) : BN;
    wrs = make_rsrc_words((const T*)p.w + (int64_t)n0 * p.w_ld, (unsigned)(((int64_t)(cols_valid - 1) * p.w_ld + p.K) * 2));   // rows past N read as zeros
    if constexpr (!XF32) {
      xrs = make_rsrc_words((const T*)p.x + (int64_t)m0 * p.x_ld, (unsigned)(((int64_t)(rows_valid - 1) * p.x_ld + p.K) * 2));
    } else {
      xtile = (const float*)p.x + (int64_t)m0 * p.x_ld;
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const int row = (tid >> 3) + 64 * j;
        const int rr = row < rows_valid ? row : rows_valid - 1;             // rows past M: copies of the last row (never stored)
        xoff[j] = (unsigned)(rr * (int)p.x_ld + (tid & 7) * 8) * 4u;
      }
    }
  };
  f32x4 xr[XF32 ? XU : 1][2];
  auto issue_w = [&](int kt, int slot) {
    unsigned char* const sx = smem + slot * STAGE;
#pragma unroll
    for (int j = 0; j < WP; ++j) buffer_load_lds16_async(wrs, sx + XT + (wid + j * NW) * 1024, wvo[j] + kt * 128);
  };
  auto issue_x = [&](int kt, int slot) {
    if constexpr (!XF32) {
      unsigned char* const sx = smem + slot * STAGE;
#pragma unroll
      for (int j = 0; j < XP; ++j) buffer_load_lds16_async(xrs, sx + (wid + j * NW) * 1024, xvo[j] + kt * 128);
    } else {
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const unsigned o = xoff[j] + (unsigned)kt * 256u;
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(xr[j][0]) : "v"(o), "s"(xtile) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(xr[j][1]) : "v"(o), "s"(xtile) : "memory");
      }
    }
  };
  auto land_stage = [&](int slot) {                     // own transfers have landed; fp32: rounded and written to LDS
    if constexpr (XF32) {
      unsigned char* const sx = smem + slot * STAGE;
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(xr[0][0]), "+v"(xr[0][1]) : : "memory");
#pragma unroll
      for (int j = 1; j < XU; ++j) asm volatile("" : "+v"(xr[j][0]), "+v"(xr[j][1]));
#pragma unroll
      for (int j = 0; j < XU; ++j) {
        const f32x8 f = __builtin_shufflevector(xr[j][0], xr[j][1], 0, 1, 2, 3, 4, 5, 6, 7);
        *(IR_LDS v8*)(IR_LDS unsigned char*)(sx + xlds0 + j * 8192) = __builtin_convertvector(f, v8);
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  f32x16 acc[MI][NI];
  v8 wf[2][NI], xf[2][MI];                              // fragments of TWO k-substeps: what one matrix phase consumes
  auto load_frags = [&](int slot, int half) {           // k-substeps 2*half, 2*half + 1 of the stage in `slot`
    const unsigned char* const st = smem + slot * STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) wf[q][ni] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(st + wrow0 + ni * 4096 + fread[2 * half + q]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[q][mi] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(st + xrow0 + mi * 4096 + fread[2 * half + q]);
    }
  };
  auto matrix_phase = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Tr::mfma(wf[q][ni], xf[q][mi], acc[mi][ni]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto phase_end = [&]() {   // fragment reads and LDS writes of this phase have left LDS; then the workgroup-wide rendezvous
    __builtin_amdgcn_sched_barrier(0);      // nothing (MFMAs are register-only: hipcc would float them) crosses a phase boundary
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // epilogue of the tile at (em0, en0): column scale, bias, ONE rounding, transposed through the wave's PRIVATE 4-KiB
  // staging area (16 rows x 144 B at a time) into whole 128-byte lines of Y.  No barrier, and nothing of it touches the
  // operand stages: the next tile's first stage lands in them meanwhile.
  auto epilogue = [&](int em0, int en0) {
    unsigned char* const tb = smem + OUT_OFF + wid * 4096;
#pragma unroll
    for (int nh = 0; nh < NI / 2; ++nh) {
      const int ncol0 = en0 + wn * (NI * 32) + nh * 64;
      if (ncol0 >= p.N) break;                            // whole 64-column group past N (wave-uniform)
      v4 bvs[2][4];
      if (p.bias != nullptr) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int g = 0; g < 4; ++g) bvs[n2][g] = *(const v4*)((const T*)p.bias + ncol0 + n2 * 32 + 8 * g + 4 * hi);
      } else {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int g = 0; g < 4; ++g) bvs[n2][g] = v4{0, 0, 0, 0};
      }
      // round 4: token statistics of a V head from the finished rows on their way out (ir_colstats.h; see the other kernel)
      const int st_head = (ncol0 - p.st_col0) >> 6;
      const bool st_on = p.st_ws != nullptr && ncol0 >= p.st_col0 && st_head < (p.st_cols >> 6) && (em0 + wm * (MI * 32)) < p.M;
      ColStatsAcc sacc;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if ((lq >> 4) == half) {
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
              const float cs = (ncol0 + n2 * 32) < p.scale_cols ? p.col_scale : 1.0f;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 f;
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = acc[mi][2 * nh + n2][4 * g + i] * cs + (float)bvs[n2][g][i];
                *(v4*)(tb + (lq & 15) * kTiledPitch + n2 * 64 + (8 * g + 4 * hi) * 2) = __builtin_convertvector(f, v4);
              }
            }
          }
          ir_wave_lds_fence();   // every lane reads rows that the sixteen staging lanes have just written (ir_common.h)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int r = 8 * j + (lane >> 3);
            const u32x4 v = *(const u32x4_alias*)(tb + r * kTiledPitch + (lane & 7) * 16);
            const int row = em0 + wm * (MI * 32) + mi * 32 + half * 16 + r;
            if (row < p.M) ir_store_y((u32x4*)((T*)p.y + (int64_t)row * p.y_ld + ncol0 + (lane & 7) * 8), v);
            if (st_on) {
              if (mi == 0 && half == 0 && j == 0) ir_stats_first<T>(sacc, v);
              else ir_stats_add<T>(sacc, v);
            }
          }
          ir_wave_lds_fence();   // the next pass overwrites the rows just read
        }
      }
      if (st_on) ir_stats_finish(sacc, p.st_ws + ((int64_t)((em0 + wm * (MI * 32)) / kStatsRows) * (p.st_cols >> 6) + st_head) * 128);
    }
  };

  // ---- walk the tiles ---------------------------------------------------------------------------------------------------------
  if (xs >= t_count) return;                             // more workgroups than tiles on this XCD (tiny problems)
  set_tile(t_start + xs);
  issue_w(0, 0);
  issue_x(0, 0);
  for (int tl = xs; tl < t_count; tl += wg_x) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    // stage 0 of this tile is in flight (issued above / before the previous tile's epilogue)
    land_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    phase_end();
    load_frags(0, 0);                                      // L0(0)
    if (KT > 1) { issue_w(1, 1); issue_x(1, 1); }
    if (grp == 1) phase_end();                             // waves 4-7 run one phase behind
    for (int kt = 0; kt < KT; ++kt) {
      const int cur = kt & 1;
      matrix_phase();                                      // M0(kt)
      phase_end();
      load_frags(cur, 1);                                  // L1(kt); the wave's transfers of step kt+1 (issued in L0(kt)) land
      if (kt + 1 < KT) land_stage(cur ^ 1);
      phase_end();
      matrix_phase();                                      // M1(kt)
      phase_end();
      if (kt + 1 < KT) {                                   // L0(kt+1) + issue of step kt+2 into the stage last read two barriers back
        load_frags(cur ^ 1, 0);
        if (kt + 2 < KT) { issue_w(kt + 2, cur); issue_x(kt + 2, cur); }
      }
      phase_end();
    }
    if (grp == 0) phase_end();
    // every wave is behind everybody's last fragment read: the stages are free.  Next tile's first stage goes out NOW.
    const int em0 = m0, en0 = n0;
    const bool more = tl + wg_x < t_count;
    if (more) {
      set_tile(t_start + tl + wg_x);
      issue_w(0, 0);
      if constexpr (!XF32) issue_x(0, 0);                  // (fp32: its staging registers would be live across the epilogue)
    }
    epilogue(em0, en0);
    if constexpr (XF32) { if (more) issue_x(0, 0); }
  }
}

// (Round 4 also built this tile with a HALF-STAGE operand stream - stage laid out per K-half, one half issued in every load
// phase, counted waits two load phases later: four phases of flight, issue cost balanced - bit-identical results, no faster:
// profiles/r4_gemm_halfstage.txt.  Not kept.)

// row panels per walk group: every XCD owns a contiguous range of tiles ordered GM row panels x all column tiles, so the
// ~32 tiles resident on an XCD form a GM x (32 / GM) patch.  Rounds 3-4 ran 8 x 4; 4 x 8 fetches 6 % less through the L2s
// (profiles/r4_pmc_linear_tiled.txt), is ~1 % ahead per GEMM (r4_gemm_walk.txt) and 0.4 % on the two-stream step (6.809 ->
// 6.783 ms, three interleaved triples with 2 / 4 / 8): the default since the end of round 4.  IR_LIN_GM overrides it (A/B)
static int walk_group_rows() {
  static const int gm = [] { const char* e = getenv("IR_LIN_GM"); const int v = e ? atoi(e) : 0; return (v >= 1 && v <= 64) ? v : 4; }();
  return gm;
}

template <typename T, bool XF32>
hipError_t launch_pp(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  p.nsplit = walk_group_rows();
  constexpr size_t dyn = 2 * (size_t)(256 + 256) * 128 + 8 * 4096;   // two operand stages + the waves' output staging: all of LDS
  static IrOncePerDevice once;            // per instantiation (ir_common.h)
  static std::atomic<int> n_cu[64];       // compute units of each device, read once (0 = not read yet)
  int dev = -1;
  const hipError_t ea = ir_opt_in_dynamic_lds(once, (const void*)linear_tiled_pp_kernel<T, XF32>, dyn, &dev);
  if (ea != hipSuccess) return ea;
  int cus = (dev >= 0 && dev < 64) ? n_cu[dev].load(std::memory_order_relaxed) : 0;
  if (cus <= 0) {
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev < 0 ? 0 : dev) != hipSuccess || cus <= 0) cus = 256;
    if (dev >= 0 && dev < 64) n_cu[dev].store(cus, std::memory_order_relaxed);
  }
  const int MT = (p.M + 255) / 256, NTl = (p.N + 255) / 256;
  const int ntiles = MT * NTl;
  // More tiles than CUs: ONE workgroup per CU walks its share of the tiles, with the next tile's first stage in flight during
  // the epilogue (the epilogue is an HBM-write burst; a quarter of a K = 640 tile's life).  Round 3 measured this form equal to
  // one workgroup per tile and kept the latter for the dispatcher's balancing; re-measured in round 4 on the short-K shapes of
  // the step it is 1-3 % ahead per GEMM (round-4 A/B, NOTES 10.2: 102.2 -> 99.3, 33.8 -> 32.6 us) and 0.6-0.8 % on the whole
  // two-stream step (7.019 -> 6.973 ms, three interleaved pairs), 0.2-0.6 % on one stream: the default since.
  // IR_LIN_PERSISTENT=0 restores one workgroup per tile (A/B).
  static const bool persistent = [] { const char* e = getenv("IR_LIN_PERSISTENT"); return e == nullptr || e[0] != '0'; }();
  const int grid = (persistent && ntiles > cus) ? cus : ntiles;
  hipLaunchKernelGGL((linear_tiled_pp_kernel<T, XF32>), dim3((unsigned)grid), dim3(512), dyn, s, p);
  return hipGetLastError();
}

template <typename T, int WM, int WN, int MI, int NI, bool XF32, bool ILV, int KW = 1>
hipError_t launch_cfg(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  p.nsplit = walk_group_rows();
  constexpr int BM = WM * MI * 32, BN = WN * NI * 32;
  constexpr size_t stage = (size_t)(BM + BN) * 128;
  constexpr size_t epi = (size_t)WM * WN * MI * 32 * kTiledPitch;
  constexpr size_t dyn = KW * 2 * stage > epi ? KW * 2 * stage : epi;
  if (KW == 2 && ((p.K >> 6) & 1)) return hipErrorInvalidValue;   // the two K groups take whole, equal halves
  static IrOncePerDevice once;   // per instantiation (ir_common.h)
  const hipError_t ea = ir_opt_in_dynamic_lds(once, (const void*)linear_tiled_kernel<T, WM, WN, MI, NI, XF32, ILV, KW>, dyn);
  if (ea != hipSuccess) return ea;
  const int MT = (p.M + BM - 1) / BM, NTl = p.N / BN;
  hipLaunchKernelGGL((linear_tiled_kernel<T, WM, WN, MI, NI, XF32, ILV, KW>), dim3((unsigned)(MT * NTl)), dim3(WM * WN * KW * 64), dyn, s, p);
  return hipGetLastError();
}

template <typename T, bool XF32>
hipError_t launch_x(const LinearKParams& p, int cfg, hipStream_t s) {
  constexpr bool I = true;   // LDS-DMA pieces between the MFMAs (false: all at the top of the step; measured equal, +-2 %)
  switch (cfg) {
    case IR_LIN_TILE_256x128: return launch_cfg<T, 4, 2, 2, 2, XF32, I>(p, s);
    case IR_LIN_TILE_128x128: return launch_cfg<T, 2, 2, 2, 2, XF32, I>(p, s);
    case IR_LIN_TILE_128x64: return launch_cfg<T, 2, 1, 2, 2, XF32, I>(p, s);
    case IR_LIN_TILE_256x64: return launch_cfg<T, 4, 1, 2, 2, XF32, I>(p, s);
    case IR_LIN_TILE_64x128: return launch_cfg<T, 1, 2, 2, 2, XF32, I>(p, s);
    case IR_LIN_TILE_128x256: return launch_cfg<T, 2, 2, 2, 4, XF32, I>(p, s);
    case IR_LIN_TILE_128x128_K2: return launch_cfg<T, 2, 2, 2, 2, XF32, I, 2>(p, s);   // contraction split over two wave groups
    case IR_LIN_TILE_256x256: return launch_pp<T, XF32>(p, s);   // wave groups one phase apart: 1-4 % over the same tile with all
                                                                  // waves in one phase (launch_cfg<T, 4, 2, 2, 4, XF32, I>), and a ragged last column tile
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

// tile shape for a problem, from the measurements of tools/gpu_gemm_probe3.py (profiles/r3_gemm_probe*.txt): the 256x256
// tile (64 x 128 per wave: 0.75 LDS fragment reads per MFMA, half the L2 traffic per flop of 128x128; ragged last column
// tile allowed) wins as soon as its grid covers ~160 of the 256 CUs; below that 128x128 at two workgroups per CU, and
// 64-row tiles when even that grid leaves CUs idle
// Round 5: where the 128x128 grid leaves a third of the CUs idle and the contraction is long (M = 2048 rows x N = 1280, K = 1280:
// 160 tiles), the split-K form of that tile (two wave groups, half the K loop each) leads by 8-12 %
// (profiles/r5_gemm_probe_final.txt: 14.4 -> 13.2 us with 16-bit activations, 22.7 -> 19.9 with fp32 ones); with more tiles than
// CUs it loses (one 128-KiB workgroup per CU instead of two), so the rule is narrow.
int ir_linear_tiled_pick(int64_t M, int N, int K) {
  auto tiles = [&](int bm, int bn) { return ((M + bm - 1) / bm) * (int64_t)((N + bn - 1) / bn); };
  if (tiles(256, 256) >= 160) return IR_LIN_TILE_256x256;
  // (round 6: up to ONE round of 256 tiles instead of 192 - M = 1024 x N = 3840 x K = 1280, the capture q/k/v GEMM of one identity with
  //  four references: 15.0 / 21.6 us against 17.2 / 24.8 us of the plain 128 x 128 tile under graph replay, profiles/r6_gemm_probe_b1_graph.txt)
  if (N % 128 == 0 && tiles(128, 128) >= 128 && tiles(128, 128) <= 256 && K >= 1024 && ((K >> 6) & 1) == 0) return IR_LIN_TILE_128x128_K2;
  if (N % 128 == 0) return tiles(128, 128) >= 128 ? IR_LIN_TILE_128x128 : IR_LIN_TILE_64x128;
  return tiles(256, 64) >= 512 ? IR_LIN_TILE_256x64 : IR_LIN_TILE_128x64;
}

bool ir_linear_tiled_cfg_ok(int cfg, int N) {
  switch (cfg) {
    case IR_LIN_TILE_256x128: case IR_LIN_TILE_128x128: case IR_LIN_TILE_64x128: case IR_LIN_TILE_128x128_K2: return N % 128 == 0;
    case IR_LIN_TILE_128x64: case IR_LIN_TILE_256x64: return N % 64 == 0;
    case IR_LIN_TILE_128x256: return N % 256 == 0;
    case IR_LIN_TILE_256x256: return N % 64 == 0;         // ragged last column tile
    default: return false;
  }
}

hipError_t ir_launch_linear_tiled(const LinearKParams& p, int dtype, int cfg, hipStream_t s) {
  if (dtype == 1) return p.x_f32 ? launch_x<__bf16, true>(p, cfg, s) : launch_x<__bf16, false>(p, cfg, s);
  return p.x_f32 ? launch_x<_Float16, true>(p, cfg, s) : launch_x<_Float16, false>(p, cfg, s);
}
