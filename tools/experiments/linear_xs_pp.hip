// linear_xs_pp.hip - the X-stationary projection GEMM of linear_skinny.hip (Y = X W^T + bias, K = 64 * KS <= 320) with
// PING-PONG wave groups, for the row counts where X-stationary is the choice (M >= 65536: the q/k/v and out projections of
// the 64x64-token class over the reference token sets, attn_processors.py:222-230,267).
//
// Same data flow as linear_skinny_kernel: a wave keeps its 64 rows of X as MFMA B-operand fragments in registers for its
// whole life (fp32 activations rounded while loaded), W streams through LDS in 32-column chunks (LDS-DMA, swizzled source
// slots), products are issued swapped, finished chunks leave in pairs through a wave-private LDS transpose as whole
// 128-byte lines of Y.  What changes is the schedule.  There, the phases of a chunk - DMA issue, staging of the previous
// chunk's results, 40 MFMAs with the stores in between - run back to back in every wave, and the two waves of a SIMD
// (from two independent 4-wave workgroups) serialise: 2 x 2900 cycles per chunk for 2 x 1280 cycles of MFMAs (NOTES.md
// 4.5).  Here one 8-wave workgroup per CU runs a chunk as TWO barrier-separated phases per wave,
//     M(i): the 40 MFMAs of chunk i (W fragments read from LDS just ahead of them)
//     L(i): everything else - LDS-DMA issue, chunk i's results -> staging tile, the eight line stores of a finished pair
// and waves 4-7 run ONE PHASE behind waves 0-3, so every SIMD has one wave in M beside one in L.
// W ring of 3 chunks.  Chunk j is first read in M(j) of group A (slot 2j of the barrier timeline) and last in M(j) of
// group B (slot 2j+1).  Group A issues its share of chunk j at the START of L(j-2) (slot 2j-3; the ring slot's previous
// occupant j-3 was last read in slot 2j-5) and confirms it at the end of L(j-1); group B issues at the start of L(j-3)
// (slot 2j-4) and confirms at the end of M(j-1) (slot 2j-1).  A confirmation is a COUNTED wait: vector memory operations
// retire in issue order, and behind a wave's pieces of chunk j there are exactly its pieces of chunk j+1 and ONE group of
// eight line stores (pairs complete every other chunk, and stores are the last thing an L phase issues), so
// vmcnt(P + 8) leaves those in flight across the barrier.
#include <type_traits>

#include "ir_common.h"
#include "ir_kernels.h"

#ifndef XSPP_ABL
#define XSPP_ABL 0
#endif
namespace {

constexpr int NCH = 32;                     // columns of Y (rows of W) per chunk
constexpr int SUB_BYTES = NCH * 128;        // one 64-k sub-tile of a chunk: 32 rows x 128 B
constexpr int kTPitch = 144;                // staging tile row: 128 B (two chunks of a Y row) + 16 B pad

template <typename T, int KS, bool BIAS>    // K = 64 * KS
__global__ void __launch_bounds__(512, 2) linear_xs_pp_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NW = 8, NT = 512;
  constexpr int CHUNK_BYTES = KS * SUB_BYTES;
  constexpr int PIECES = 4 * KS;                       // 1-KiB LDS-DMA pieces per chunk
  constexpr int PA = (PIECES / 4 + 1) / 2;             // pieces per wave of group A (waves 0-3) ...
  constexpr int PB = PIECES / 4 - PA;                  // ... and of group B (waves 4-7): 3 + 2 at K = 320
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_xspp[];
  unsigned char* const smem = dsm_xspp;                // | 3 W chunks | 8 staging tiles | bias of the column range |
  unsigned char* const tbuf = dsm_xspp + 3 * CHUNK_BYTES;
  T* const sbias = (T*)(dsm_xspp + 3 * CHUNK_BYTES + NW * 64 * kTPitch);

#ifdef XSPP_TRACE
  const unsigned long long cal_e = __builtin_amdgcn_s_memrealtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wid >> 2, wq = wid & 3;
  const int hi = lane >> 5, lq = lane & 31;

  const int mb = blockIdx.x / p.nsplit, sp = blockIdx.x - mb * p.nsplit;
  const int nchunks = p.N / NCH;
  const int unit = (nchunks & 1) ? 1 : 2, nu = nchunks / unit;   // ranges start on even chunks when they can (whole-line stores)
  const int c_begin = unit * (int)(((long)nu * sp) / p.nsplit), c_end = unit * (int)(((long)nu * (sp + 1)) / p.nsplit);
  if (c_begin >= c_end) return;
  const int ncl = c_end - c_begin;

  // ---- X fragments of both 32-row blocks: resident for the whole kernel ------------------------------------------------------
  const int rowA = mb * NT + wid * 64 + lq, rowB = rowA + 32;
  v8 xA[4 * KS], xB[4 * KS];
  {
    const int ra = rowA < p.M ? rowA : p.M - 1, rb = rowB < p.M ? rowB : p.M - 1;
    if (p.x_f32) {   // one row block at a time: all 8 * KS fp32 fragments in flight at once would spill resident ones
      const float* base = (const float*)p.x + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4 * KS; ++ks)
        xA[ks] = __builtin_convertvector(*(const f32x8*)(base + (int64_t)ra * p.x_ld + ks * 16), v8);
#pragma unroll
      for (int ks = 0; ks < 4 * KS; ++ks) asm volatile("" : "+v"(xA[ks]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4 * KS; ++ks)
        xB[ks] = __builtin_convertvector(*(const f32x8*)(base + (int64_t)rb * p.x_ld + ks * 16), v8);
    } else {
      const T* base = (const T*)p.x + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4 * KS; ++ks) {
        xA[ks] = *(const v8*)(base + (int64_t)ra * p.x_ld + ks * 16);
        xB[ks] = *(const v8*)(base + (int64_t)rb * p.x_ld + ks * 16);
      }
    }
  }

  // ---- W chunk stream: piece q = 4 * s + (quarter of sub-tile s); group A waves take q = wq, wq + 4, ..., group B the rest -------
  const i32x4 wrw = make_rsrc_words(p.w, (unsigned)(((int64_t)(p.N - 1) * p.w_ld + 64 * KS) * 2));
  // a piece = 8 rows x 128 B of a sub-tile: lane -> (row of the piece, 16-B slot), swizzle on the SOURCE slot
  const int prow = lane >> 3, pslot = lane & 7;
  auto piece_off = [&](int quarter) {   // offset of this lane's 16 bytes inside a chunk's global image, sub-tile 0
    const int row = quarter * 8 + prow;
    return (unsigned)(row * p.w_ld * 2 + ((pslot ^ ((row >> 1) & 7)) * 16));
  };
  const unsigned po = piece_off(wq);
  auto issue_chunk = [&](int c, int slot) {             // this wave's pieces of chunk c (global chunk index) into ring slot `slot`
    const unsigned off = po + (unsigned)((int64_t)c * NCH * p.w_ld * 2);
    unsigned char* const dst = smem + slot * CHUNK_BYTES + wq * 1024;
    if (grp == 0) {
#pragma unroll
      for (int j = 0; j < PA; ++j) buffer_load_lds16_async(wrw, dst + j * SUB_BYTES, off + j * 128);          // sub-tiles 0 .. PA-1
    } else {
#pragma unroll
      for (int j = 0; j < PB; ++j) buffer_load_lds16_async(wrw, dst + (PA + j) * SUB_BYTES, off + (PA + j) * 128);   // sub-tiles PA .. KS-1
    }
  };
  int wread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wread[ks] = lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);

  if (BIAS)
    for (int i = tid; i < ncl * NCH; i += NT) sbias[i] = ((const T*)p.bias)[c_begin * NCH + i];

  unsigned char* const tb = tbuf + wid * (64 * kTPitch);
  auto stage_block = [&](const f32x16& acc, int rbase, int n0, int half) {
    const float cs = n0 < p.scale_cols ? p.col_scale : 1.0f;   // leading columns scaled in fp32 before the one rounding
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 f;
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = acc[4 * g + i] * cs;
      if (BIAS) {
        const v4 bv = *(const v4*)(sbias + (n0 - c_begin * NCH) + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] += (float)bv[i];
      }
      *(v4*)(tb + (rbase + lq) * kTPitch + half * 64 + (8 * g + 4 * hi) * 2) = __builtin_convertvector(f, v4);
    }
  };
  // Y leaves through buffer stores: ONE per-lane byte offset for all of them, the row-group / column part in an SGPR, and
  // rows past M dropped by the descriptor's range check (the store is still issued: the count per phase stays constant)
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(unsigned)((int64_t)p.M * p.y_ld * 2), 0x00020000);
  const int row0 = mb * NT + wid * 64;
  const int yvo_full = ((row0 + (lane >> 3)) * (int)p.y_ld + (lane & 7) * 8) * 2;
  const int yvo_half = ((row0 + (lane >> 2)) * (int)p.y_ld + (lane & 3) * 8) * 2;
  const int ystep = (int)p.y_ld * 16;                   // bytes between row groups of a full-line store (8 rows)
  auto store_full = [&](int j, int n0) {   // 8 rows x 128 B: one of the eight stores of a finished chunk PAIR
    const u32x4 v = *(const u32x4*)(tb + (8 * j + (lane >> 3)) * kTPitch + (lane & 7) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, yrs, yvo_full, __builtin_amdgcn_readfirstlane(j * ystep + n0 * 2), 0);
  };
  auto store_half = [&](int j, int n0) {   // 16 rows x 64 B (left half of the tile): a range's odd last chunk
    const u32x4 v = *(const u32x4*)(tb + (16 * j + (lane >> 2)) * kTPitch + (lane & 3) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, yrs, yvo_half, __builtin_amdgcn_readfirstlane(2 * j * ystep + n0 * 2), 0);
  };
#ifdef XSPP_TRACE
  unsigned long long* const trc = (unsigned long long*)(dsm_xspp + 3 * CHUNK_BYTES + NW * 64 * kTPitch + 8192) + (wid >> 2) * 256;
  int trn = 0;
  const bool tron = (blockIdx.x == XSPP_TRACE) && wq == 0;
  auto stamp = [&]() {
    if (tron) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0 && trn < 256) trc[trn] = t;
      ++trn;
    }
  };
#else
  auto stamp = [&]() {};
#endif
#ifdef XSPP_TRACE_L
  auto stamp2 = [&]() { __builtin_amdgcn_sched_barrier(0); stamp(); __builtin_amdgcn_sched_barrier(0); };
#else
  auto stamp2 = [&]() {};
#endif
  auto phase_end = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // A wave's pieces of chunk i+1 have landed when at most (its pieces of chunk i+2) + (the stores of the two L phases since)
  // are outstanding: 4 stores per L phase from L(1) on, `k` = how many of those two phases had them
  auto confirm = [&](bool more, int k) {
    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (grp == 0) {
      if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + 8) : "memory");
      else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA) : "memory");
    } else {
      if (k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB + 8) : "memory");
      else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB + 4) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory");
    }
  };

  // ---- prologue: chunks 0, 1 (and group B's share of chunk 2) on their way, X resident ----------------------------------------
  issue_chunk(c_begin, 0);
  if (ncl > 1) issue_chunk(c_begin + 1, 1);
  if (grp == 1 && ncl > 2) issue_chunk(c_begin + 2, 2);
#pragma unroll
  for (int ks = 0; ks < 4 * KS; ++ks) asm volatile("" ::"v"(xA[ks]), "v"(xB[ks]));  // the X loads retire before the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                       // (also publishes sbias)
  if (grp == 1) phase_end();                             // waves 4-7 run one phase behind

#ifdef XSPP_TRACE
  const unsigned long long cal_m0 = __builtin_amdgcn_s_memtime(), cal_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  f32x16 accA, accB;
  int ring = 0;                                          // ring slot of chunk i
  for (int i = 0; i < ncl; ++i) {
    const int c = c_begin + i;
    stamp();
    // ---- M(i): 40 MFMAs; W fragments double-buffered by 64-k sub-tile ------------------------------------------------------------
    {
      const unsigned char* Wb = smem + ring * CHUNK_BYTES;
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
      v8 wf[2][4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[0][ks] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + wread[ks]);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
            wf[(s + 1) & 1][ks] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + (s + 1) * SUB_BYTES + wread[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (!(XSPP_ABL & 1)) {
            accA = Tr::mfma(wf[s & 1][ks], xA[4 * s + ks], accA);
            accB = Tr::mfma(wf[s & 1][ks], xB[4 * s + ks], accB);
          } else {
            asm volatile("" ::"v"(wf[s & 1][ks]));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_setprio(0);
      stamp();
      if (grp == 1) confirm(i + 2 < ncl, i - 1);   // group B confirms its pieces of chunk i+1 here, one barrier ahead of group A's M(i+1)
    }
    stamp();
    phase_end();
    stamp();
    // ---- L(i): DMA issue first, then chunk i's results into the staging tile, then (pair complete) its eight line stores ----------
    {
      const int ahead = grp == 0 ? 2 : 3;
      if (i + ahead < ncl && !(XSPP_ABL & 4)) {
        int slot = ring + ahead;                          // (i + ahead) % 3
        slot = slot >= 3 ? slot - 3 : slot;
        slot = slot >= 3 ? slot - 3 : slot;
        issue_chunk(c + ahead, slot);
      }
      __builtin_amdgcn_sched_barrier(0);                  // the counted waits below rely on the pieces going out ahead of the stores
      stamp2();
      // the eight line stores of a finished pair leave in two halves, rows 0-31 in the L phase that completes the pair and
      // rows 32-63 at the start of the next one (ahead of the staging that reuses the tile's left half): every L phase
      // issues four stores, and the chip sees a steady write stream instead of all CUs bursting every other chunk
      if (i > 0 && !(i & 1) && !(XSPP_ABL & 2)) {
#pragma unroll
        for (int j = 4; j < 8; ++j) store_full(j, (c - 2) * NCH);
      }
      stamp2();
      const bool last = (i + 1 == ncl);
      const bool odd_tail = last && (ncl & 1);            // a range's odd last chunk leaves alone, in half lines
      if (!(XSPP_ABL & 8)) {
        stage_block(accA, 0, c * NCH, odd_tail ? 0 : (i & 1));
        stage_block(accB, 32, c * NCH, odd_tail ? 0 : (i & 1));
      } else {
        asm volatile("" ::"v"(accA), "v"(accB));
      }
      stamp2();
      if (XSPP_ABL & 2) {
      } else if (odd_tail) {
#pragma unroll
        for (int j = 0; j < 4; ++j) store_half(j, c * NCH);
      } else if (i & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) store_full(j, (c - 1) * NCH);
      }
      stamp();
      if (grp == 0) confirm(i + 2 < ncl, i);   // group A confirms its pieces of chunk i+1 (issued at the start of L(i-1))
    }
    stamp();
    phase_end();
    ring = ring == 2 ? 0 : ring + 1;
  }
  if (!(ncl & 1) && !(XSPP_ABL & 2)) {                 // rows 32-63 of the last pair
#pragma unroll
    for (int j = 4; j < 8; ++j) store_full(j, (c_end - 2) * NCH);
  }
  if (grp == 0) phase_end();
#ifdef XSPP_TRACE
  if (tron && lane == 0 && grp == 0) {
    trc[252] = cal_m0; trc[253] = __builtin_amdgcn_s_memtime(); trc[254] = cal_r0; trc[255] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  if (blockIdx.x != XSPP_TRACE && tid == 0) {      // every other block: entry | loop start | loop end (100 MHz ticks) | XCC_ID, over its first row of Y
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* o = (unsigned long long*)((T*)p.y + (int64_t)row0 * p.y_ld);
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    o[0] = cal_e; o[1] = cal_r0; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = xcc;
  }
  if (blockIdx.x == XSPP_TRACE && tid < 512) {   // both groups' stamps over the first bytes of Y (debug build only)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ((unsigned long long*)p.y)[tid] = ((unsigned long long*)(dsm_xspp + 3 * CHUNK_BYTES + NW * 64 * kTPitch + 8192))[tid];
  }
#endif
}

template <typename T, int KS, bool BIAS>
hipError_t launch_pp(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  const int mblocks = (p.M + 511) / 512;
  const int nchunks = p.N / NCH;
  int nsplit = (256 + mblocks - 1) / mblocks;                // one 8-wave workgroup per CU
  const int nunits = (nchunks & 1) ? nchunks : nchunks / 2;
  if (nsplit > nunits) nsplit = nunits;
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
#ifdef XSPP_TRACE
  const size_t dyn = (size_t)3 * KS * SUB_BYTES + (size_t)8 * 64 * kTPitch + 8192 + 4096;
#else
  const size_t dyn = (size_t)3 * KS * SUB_BYTES + (size_t)8 * 64 * kTPitch + (BIAS ? kLinearMaxBiasN * sizeof(T) : 0);
#endif
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, attr_set[0] = false;
  if (!attr_set[dev]) {
    hipError_t ea = hipFuncSetAttribute((const void*)linear_xs_pp_kernel<T, KS, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (ea != hipSuccess) return ea;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((linear_xs_pp_kernel<T, KS, BIAS>), dim3((unsigned)(mblocks * nsplit)), dim3(512), dyn, s, p);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_t(const LinearKParams& p, hipStream_t s) {
  if (p.K != 320) return hipErrorInvalidValue;
  return p.bias != nullptr ? launch_pp<T, 5, true>(p, s) : launch_pp<T, 5, false>(p, s);
}

}  // namespace

bool ir_linear_xs_pp_covers(int N, int K, bool has_bias) {
  return K == 320 && N % NCH == 0 && (!has_bias || N <= kLinearMaxBiasN);
}

hipError_t ir_launch_linear_xs_pp(const LinearKParams& p, int dtype, hipStream_t s) {
  return dtype == 1 ? launch_t<__bf16>(p, s) : launch_t<_Float16>(p, s);
}
