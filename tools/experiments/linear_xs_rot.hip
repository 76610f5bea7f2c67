// linear_xs_rot.hip - the X-stationary projection GEMM of linear_skinny.hip (Y = X W^T + bias, K = 320) with the two 32-row
// blocks of a wave ROTATED against each other, so that everything that is not an MFMA rides in the shadow of one.
//
// Same data flow as linear_skinny_kernel: a wave keeps its 64 rows of X as MFMA B-operand fragments in registers for its
// whole life (fp32 activations rounded while loaded), W streams through LDS in 32-column chunks (LDS-DMA, swizzled source
// slots, a ring of three), products are issued swapped, finished chunks leave in pairs through a wave-private LDS
// transpose as whole 128-byte lines of Y.  There, a chunk is: stage the previous chunk's 64 x 32 results (they sit in the
// accumulators the next MFMAs overwrite, so the staging cannot overlap them), then 40 MFMAs alternating between the two
// row blocks; measured (phase stamps, tools/_xspp_trace.py) the staging alone costs as much as two thirds of the MFMAs.
// Here a chunk is two WINDOWS of 20 MFMAs, one per row block:
//     window A(i): accA <- W(i) . xA     while the wave stages accB of chunk i-1 and stores finished lines
//     window B(i): accB <- W(i) . xB     while it stages accA of chunk i and stores finished lines
// An MFMA of this shape keeps the matrix pipe busy for 8 passes; the wave cannot issue its next one sooner, and the
// staging / store instructions between two MFMAs fill issue slots that were idle.  No second accumulator set is needed.
// The price: every W fragment is read from LDS twice (once per window), 320 KB per chunk and CU at 256 B/clk.
// One barrier per chunk (the W ring); a wave's pieces of chunk i+1 are confirmed with a COUNTED wait (vector memory
// operations retire in issue order: behind them are its pieces of chunk i+2 and the eight stores of ONE even chunk).
#include <type_traits>

#include "ir_common.h"
#include "ir_kernels.h"

#ifndef XSROT_PF
#define XSROT_PF 3
#endif
namespace {

constexpr int NCH = 32;                     // columns of Y (rows of W) per chunk
constexpr int SUB_BYTES = NCH * 128;        // one 64-k sub-tile of a chunk: 32 rows x 128 B
constexpr int kTPitch = 144;                // staging tile row: 128 B (two chunks of a Y row) + 16 B pad

template <typename T, int KS, bool BIAS>    // K = 64 * KS
__global__ void __launch_bounds__(512, 2) linear_xs_rot_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int NW = 8, NT = 512;
  constexpr int CHUNK_BYTES = KS * SUB_BYTES;
  constexpr int PIECES = 4 * KS;                       // 1-KiB LDS-DMA pieces per chunk
  constexpr int PA = (PIECES / 4 + 1) / 2;             // pieces per wave of group A (waves 0-3) ...
  constexpr int PB = PIECES / 4 - PA;                  // ... and of group B (waves 4-7): 3 + 2 at K = 320
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_xsrot[];
  unsigned char* const smem = dsm_xsrot;                // | 3 W chunks | 8 staging tiles | bias of the column range |
  unsigned char* const tbuf = dsm_xsrot + 3 * CHUNK_BYTES;
  T* const sbias = (T*)(dsm_xsrot + 3 * CHUNK_BYTES + NW * 64 * kTPitch);

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
  // one quarter (accumulator registers 4g .. 4g+3 = columns 8g + 4hi .. +3 of the lane's row) of a finished 32 x 32 block
  // into the wave's staging tile: column scale and bias in fp32, ONE rounding
  auto stage_piece = [&](const f32x16& acc, int rbase, int n0, int half, int g) {
    const float cs = n0 < p.scale_cols ? p.col_scale : 1.0f;   // leading columns scaled in fp32 before the one rounding
    f32x4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = acc[4 * g + i] * cs;
    if (BIAS) {
      const v4 bv = *(const v4*)(sbias + (n0 - c_begin * NCH) + 8 * g + 4 * hi);
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] += (float)bv[i];
    }
    *(v4*)(tb + (rbase + lq) * kTPitch + half * 64 + (8 * g + 4 * hi) * 2) = __builtin_convertvector(f, v4);
  };
  // Y leaves through buffer stores: ONE per-lane byte offset for all of them, the row-group / column part in an SGPR, and
  // rows past M dropped by the descriptor's range check (the store is still issued: the count per phase stays constant)
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)(unsigned)((int64_t)p.M * p.y_ld * 2), 0x00020000);
  const int row0 = mb * NT + wid * 64;
  const int yvo_full = ((row0 + (lane >> 3)) * (int)p.y_ld + (lane & 7) * 8) * 2;
  const int yvo_half = ((row0 + (lane >> 2)) * (int)p.y_ld + (lane & 3) * 8) * 2;
  const int ystep = (int)p.y_ld * 16;                   // bytes between row groups of a full-line store (8 rows)
  // 8 rows x 128 B: one of the eight stores of a finished chunk PAIR, as a tile read and (an MFMA or two later) the store
  auto line_read = [&](int j) { return *(const u32x4*)(tb + (8 * j + (lane >> 3)) * kTPitch + (lane & 7) * 16); };
  auto line_store = [&](const u32x4& v, int j, int n0) {
    __builtin_amdgcn_raw_buffer_store_b128(v, yrs, yvo_full, __builtin_amdgcn_readfirstlane(j * ystep + n0 * 2), 0);
  };
  auto store_half = [&](int j, int n0) {   // 16 rows x 64 B (left half of the tile): a range's odd last chunk
    const u32x4 v = *(const u32x4*)(tb + (16 * j + (lane >> 2)) * kTPitch + (lane & 3) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, yrs, yvo_half, __builtin_amdgcn_readfirstlane(2 * j * ystep + n0 * 2), 0);
  };

  // ---- a window: the 20 MFMAs of one row block over one chunk, W fragments PF ahead; `kind`: what rides between them -------
  //   0 nothing | 1 one block staged | 2 one block staged, then four line stores | 3 four line stores, then one block staged
  const f32x16 zero = {};
  auto window = [&](auto kind, f32x16& acc, const v8 (&x)[4 * KS], const unsigned char* Wb, const f32x16& sacc, int srbase,
                    int sn0, int shalf, int jbase, int stn0) {
    constexpr int KIND = decltype(kind)::value;
    constexpr int NM = 4 * KS, PF = XSROT_PF;
    auto wload = [&](int t) { return *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + (t >> 2) * SUB_BYTES + wread[t & 3]); };
    v8 wf[PF + 1];
    u32x4 sv[2];
#pragma unroll
    for (int t = 0; t < PF; ++t) wf[t] = wload(t);
#pragma unroll
    for (int t = 0; t < NM; ++t) {
      if (t + PF < NM) wf[(t + PF) % (PF + 1)] = wload(t + PF);
      acc = Tr::mfma(wf[t % (PF + 1)], x[t], t == 0 ? zero : acc);
      if (KIND == 1) {
        if ((t & 3) == 3 && t < 16) stage_piece(sacc, srbase, sn0, shalf, t >> 2);
      } else if (KIND == 2 || KIND == 3) {
        const int ts = KIND == 2 ? t : t - 8;            // the staging quarter of the window ...
        const int tl = KIND == 2 ? t - 8 : t;            // ... and the store quarter
        if (ts >= 1 && ts < 9 && (ts & 1)) stage_piece(sacc, srbase, sn0, shalf, ts >> 1);
        if (tl == 0) sv[0] = line_read(jbase);
        if (tl == 1) sv[1] = line_read(jbase + 1);
        if (tl == 2) { line_store(sv[0], jbase, stn0); sv[0] = line_read(jbase + 2); }
        if (tl == 3) { line_store(sv[1], jbase + 1, stn0); sv[1] = line_read(jbase + 3); }
        if (tl == 5) line_store(sv[0], jbase + 2, stn0);
        if (tl == 6) line_store(sv[1], jbase + 3, stn0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

#ifdef XSROT_TRACE
  unsigned long long* const trc = (unsigned long long*)(dsm_xsrot + 3 * CHUNK_BYTES + NW * 64 * kTPitch + 8192) + (wid >> 2) * 256;
  int trn = 0;
  const bool tron = (blockIdx.x == XSROT_TRACE) && wq == 0;
  auto stamp = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    if (tron) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0 && trn < 250) trc[trn] = t;
      ++trn;
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#else
  auto stamp = [&]() {};
#endif
  // ---- prologue: chunks 0 and 1 on their way, X resident -------------------------------------------------------------------
  issue_chunk(c_begin, 0);
  if (ncl > 1) issue_chunk(c_begin + 1, 1);
#pragma unroll
  for (int ks = 0; ks < 4 * KS; ++ks) asm volatile("" ::"v"(xA[ks]), "v"(xB[ks]));  // the X loads retire before the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                       // (also publishes sbias)

#ifdef XSROT_TRACE
  const unsigned long long cal_m0 = __builtin_amdgcn_s_memtime(), cal_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  f32x16 accA = zero, accB = zero;
  int ring = 0;                                          // ring slot of chunk i
  for (int i = 0; i < ncl; ++i) {
    const int c = c_begin + i;
    const unsigned char* Wb = smem + ring * CHUNK_BYTES;
    stamp();
    if (i + 2 < ncl) issue_chunk(c + 2, ring == 0 ? 2 : ring - 1);   // into the slot of chunk i-1: every wave is past the barrier behind it
    __builtin_amdgcn_sched_barrier(0);                   // the counted wait below relies on the pieces going out ahead of the stores
    if (i == 0) {
      window(K0{}, accA, xA, Wb, accB, 0, 0, 0, 0, 0);
      stamp();
      window(K1{}, accB, xB, Wb, accA, 0, c * NCH, 0, 0, 0);
    } else if (i & 1) {            // an odd chunk completes a pair: its blocks go to the right half of the tile
      window(K1{}, accA, xA, Wb, accB, 32, (c - 1) * NCH, 0, 0, 0);
      stamp();
      window(K1{}, accB, xB, Wb, accA, 0, c * NCH, 1, 0, 0);
    } else {                       // an even chunk stores the pair before it: rows 0-31 beside the staging of rows 32-63, then rows 32-63 ahead of the staging that reuses rows 0-31
      window(K2{}, accA, xA, Wb, accB, 32, (c - 1) * NCH, 1, 0, (c - 2) * NCH);
      stamp();
      window(K3{}, accB, xB, Wb, accA, 0, c * NCH, 0, 4, (c - 2) * NCH);
    }
    stamp();
    // this wave's pieces of chunk i+1 have landed when at most (its pieces of chunk i+2) + (the eight stores of the even
    // chunk among i-1, i) are outstanding
    if (i + 1 < ncl) {
      const bool more = i + 2 < ncl, st = i >= 2;
      if (grp == 0) {
        if (more && st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + 8) : "memory");
        else if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA) : "memory");
        else if (st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        if (more && st) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB + 8) : "memory");
        else if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory");
        else if (st) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      stamp();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    ring = ring == 2 ? 0 : ring + 1;
  }
  // ---- rows 32-63 of the last chunk, then the last pair (or a range's odd last chunk, alone, in half lines) -------------------------
  const int lc = c_end - 1;
#pragma unroll
  for (int g = 0; g < 4; ++g) stage_piece(accB, 32, lc * NCH, (ncl & 1) ? 0 : 1, g);
  if (ncl & 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) store_half(j, lc * NCH);
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) line_store(line_read(j), j, (lc - 1) * NCH);
  }
#ifdef XSROT_TRACE
  if (tron && lane == 0 && grp == 0) {
    trc[252] = cal_m0; trc[253] = __builtin_amdgcn_s_memtime(); trc[254] = cal_r0; trc[255] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  if (blockIdx.x == XSROT_TRACE) {   // both wave groups' stamps over the first bytes of Y (debug build only)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ((unsigned long long*)p.y)[tid] = ((unsigned long long*)(dsm_xsrot + 3 * CHUNK_BYTES + NW * 64 * kTPitch + 8192))[tid];
  }
#endif
}

template <typename T, int KS, bool BIAS>
hipError_t launch_rot(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  const int mblocks = (p.M + 511) / 512;
  const int nchunks = p.N / NCH;
  int nsplit = (256 + mblocks - 1) / mblocks;                // one 8-wave workgroup per CU
  const int nunits = (nchunks & 1) ? nchunks : nchunks / 2;
  if (nsplit > nunits) nsplit = nunits;
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
#ifdef XSROT_TRACE
  const size_t dyn = (size_t)3 * KS * SUB_BYTES + (size_t)8 * 64 * kTPitch + 8192 + 4096;
#else
  const size_t dyn = (size_t)3 * KS * SUB_BYTES + (size_t)8 * 64 * kTPitch + (BIAS ? kLinearMaxBiasN * sizeof(T) : 0);
#endif
  static bool attr_set[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0, attr_set[0] = false;
  if (!attr_set[dev]) {
    hipError_t ea = hipFuncSetAttribute((const void*)linear_xs_rot_kernel<T, KS, BIAS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    if (ea != hipSuccess) return ea;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL((linear_xs_rot_kernel<T, KS, BIAS>), dim3((unsigned)(mblocks * nsplit)), dim3(512), dyn, s, p);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_t(const LinearKParams& p, hipStream_t s) {
  if (p.K != 320) return hipErrorInvalidValue;
  return p.bias != nullptr ? launch_rot<T, 5, true>(p, s) : launch_rot<T, 5, false>(p, s);
}

}  // namespace

bool ir_linear_xs_rot_covers(int N, int K, bool has_bias) {
  return K == 320 && N % NCH == 0 && (!has_bias || N <= kLinearMaxBiasN);
}

hipError_t ir_launch_linear_xs_rot(const LinearKParams& p, int dtype, hipStream_t s) {
  return dtype == 1 ? launch_t<__bf16>(p, s) : launch_t<_Float16>(p, s);
}
