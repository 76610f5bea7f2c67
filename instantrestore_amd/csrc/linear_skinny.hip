// linear_skinny.hip - Y[M,N] = X[M,K] W[N,K]^T (+ bias) for the projections of the 64x64-token layer
// class (K = C = 320: to_q/to_k/to_v fused into one N = 960 GEMM, to_out with N = 320), gfx950.
//
// Why not the vendor GEMM here (it stays the path for every other shape): at K = 320 the problem is
// a streaming one - 2*(M*K + M*N) bytes against 2*M*N*K flops puts it under the ridge - and a
// 256x256x64 macro-tile kernel spends its five K-slabs on prologue/epilogue and re-reads X once per
// column tile: hipBLASLt reaches ~30 % of the HBM roofline on (131072 x 960 x 320)
// (tools/gpu_gemm_probe.py).  This kernel is X-stationary: a wave keeps its 64 rows of X as MFMA
// B-operand fragments in registers for its whole life (K/16 fragments per 32-row block) and streams
// W through LDS in 32-column chunks (LDS-DMA, the attention kernels' K-tile swizzle), so X is read
// once, W comes from L2, and Y leaves as 16-byte stores.  Both products are issued "swapped"
// (Y^T = W X^T) like the attention kernels: a lane owns one row of Y, its registers walk along n.
#include <type_traits>

#include "ir_common.h"
#include "ir_colstats.h"
#include "ir_kernels.h"

namespace {

constexpr int NCH = 32;                     // columns of Y (rows of W) per chunk
constexpr int SUB_BYTES = NCH * 128;        // one 64-k sub-tile of a chunk: 32 rows x 128 B
constexpr int kSkinnyTPitch = 144;          // staging tile row: 128 B (two chunks of a Y row) + 16 B pad

// NW waves of 64 rows share every W chunk: 4 (two workgroups per CU) or 8 (one; half the LDS-DMA pieces per wave and
// half the W traffic from L2 - the form for row counts that fill the chip with 512-row workgroups).  BIAS: compile-time,
// so that the staging code of the bias-free projections carries no branches.
template <typename T, int KS, bool BIAS, int NW>   // K = 64 * KS
__global__ void __launch_bounds__(NW * 64, 2) linear_skinny_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int CHUNK_BYTES = KS * SUB_BYTES;
  // dynamic LDS: two W chunks (one being read, one in flight) | per-wave output staging tiles | bias of this column range
  // (a global bias load inside the loop would make hipcc wait on the in-flight W chunk as well)
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm_skinny[];
  unsigned char* const smem = dsm_skinny;
  constexpr int TPITCH = kSkinnyTPitch;      // bytes per row of a staging tile: TWO chunks = 128 B of a Y row, + pad
  unsigned char* const tbuf = dsm_skinny + 2 * CHUNK_BYTES;
  T* const sbias = (T*)(dsm_skinny + 2 * CHUNK_BYTES + NW * 64 * TPITCH);
  constexpr int NT = NW * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lq = lane & 31;

  // ---- work decode: (row block of 256, range of column chunks) -------------------------------------
  const int mb = blockIdx.x / p.nsplit, sp = blockIdx.x - mb * p.nsplit;
  const int nchunks = p.N / NCH;
  // column ranges start on EVEN chunks when they can: finished chunks leave in pairs, as whole 128-B lines of Y
  const int unit = (nchunks & 1) ? 1 : 2, nu = nchunks / unit;
  const int c_begin = unit * (int)(((long)nu * sp) / p.nsplit), c_end = unit * (int)(((long)nu * (sp + 1)) / p.nsplit);
  if (c_begin >= c_end) return;

  // ---- X fragments of both 32-row blocks: resident for the whole kernel ------------------------------
  const int rowA = mb * NT + wid * 64 + lq, rowB = rowA + 32;
  v8 xA[4 * KS], xB[4 * KS];
  {
    const int ra = rowA < p.M ? rowA : p.M - 1, rb = rowB < p.M ? rowB : p.M - 1;
    if (p.x_f32) {
      // fp32 activations (LayerNorm output under autocast, test.py:83): the cast to the 16-bit compute type happens
      // here, on the way into the resident fragments (round to nearest even, the bytes `.to(dtype)` would produce) -
      // instead of a separate pass that reads 4 and writes 2 bytes per element ahead of every projection
      // one row block at a time: all 40 fp32 fragments in flight at once would need 320 registers and the allocator
      // answers by spilling resident fragments to scratch (reloaded in the main loop, each reload draining the W stream)
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

  // ---- W chunk stream: 16 B per thread per sub-tile, lane-linear LDS image, swizzle on the source ------
  // a sub-tile (32 rows x 128 B) is four 1-KiB pieces; wave w moves piece (w & 3) of the sub-tiles s = (w >> 2), (w >> 2) + NW/4, ...
  const int t2 = tid & 255, wq = wid & 3, wh = wid >> 2;   // wave-uniform: the LDS destination of a transfer lives in M0
  const int wrow = t2 >> 3, wslot = t2 & 7;
  const i32x4 wrw = make_rsrc_words(p.w, (unsigned)(((int64_t)(p.N - 1) * p.w_ld + 64 * KS) * 2));
  const unsigned wvo = (unsigned)(wrow * p.w_ld * 2 + ((wslot ^ ((wrow >> 1) & 7)) * 16));
  auto issue_chunk = [&](int c, int slot) {
    const unsigned off = wvo + (unsigned)((int64_t)c * NCH * p.w_ld * 2);
#pragma unroll
    for (int j = 0; j < (KS + NW / 4 - 1) / (NW / 4); ++j) {
      const int s = j * (NW / 4) + wh;
      if (s < KS) buffer_load_lds16_async(wrw, smem + slot * CHUNK_BYTES + s * SUB_BYTES + wq * 1024, off + s * 128);
    }
  };
  // fragment ks of a sub-tile sits at lq * 128 + (((2 ks + hi) ^ swz) << 4) = wread0 ^ (ks << 5): ONE register, re-derived
  // per chunk behind an opaque statement - four hoisted addresses were the registers that tipped the K = 320 kernel into a
  // scratch reload inside this loop (and every scratch reload waits for vmcnt(0): the Y stores in flight)
  int wread0 = lq * 128 + ((hi ^ ((lq >> 1) & 7)) << 4);

  if (BIAS)
    for (int i = tid; i < (c_end - c_begin) * NCH; i += NT) sbias[i] = ((const T*)p.bias)[c_begin * NCH + i];
  const int ncl = c_end - c_begin;
  issue_chunk(c_begin, 0);
#pragma unroll
  for (int ks = 0; ks < 4 * KS; ++ks) asm volatile("" ::"v"(xA[ks]), "v"(xB[ks]));  // X loads retire before the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // A finished chunk (64 rows x 32 columns per wave) leaves through a wave-private LDS tile: in the MFMA layout a lane
  // owns 4-column groups of ONE row, so direct stores touch 32 rows with 16-32 B each and the write path drowns in
  // partial-line requests.  The tile holds TWO consecutive chunks side by side (64 columns = 128 B per row): after the
  // transpose eight neighbouring lanes write one whole 128-B line of Y, 8 rows per store instruction.  (Round 2: with
  // 64-B half lines per chunk the Y stream ran at 2.3 TB/s - each line was written in two halves an iteration apart.)
  unsigned char* tb = tbuf + wid * (64 * TPITCH);
  auto stage_block = [&](const f32x16& acc, int rbase, int n0, int half) {
    ir_wave_lds_fence();   // these writes land on rows other lanes may just have read (ir_common.h)
    const float cs = n0 < p.scale_cols ? p.col_scale : 1.0f;   // leading columns scaled in fp32 before the one rounding
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 f;
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = acc[4 * g + i] * cs;
      if (BIAS && n0 < p.scale_cols) {   // scaled columns: (sum) * cs + bias; everywhere else the accumulators STARTED at the bias
        const v4 bv = *(const v4*)(sbias + (n0 - c_begin * NCH) + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] += (float)bv[i];
      }
      *(v4*)(tb + (rbase + lq) * TPITCH + half * 64 + (8 * g + 4 * hi) * 2) = __builtin_convertvector(f, v4);
    }
  };
  // Stores go through a buffer resource over THIS wave's rows of Y: one per-lane byte offset (row r of 8 j + (lane >> 3),
  // 16-byte column slot lane & 7) serves every store - the j and column terms are wave-uniform and ride in the scalar
  // offset - and rows past M fall outside num_records, where the hardware drops them.  (Rounds 1-3 held eight clamped
  // 64-bit row addresses in sixteen registers across the main loop; with them the K = 320 kernels spilled X fragments.)
  // Dropped or not, a store counts in vmcnt: the s_waitcnt arithmetic of the main loop relies on eight per chunk pair.
  const int row0 = mb * NT + wid * 64;
  const int rows_here = (p.M - row0) < 64 ? (p.M - row0) : 64;   // >= 1: the grid covers ceil(M / NT) blocks, a wave past M has none
  const i32x4 yrw = make_rsrc_words((const T*)p.y + (int64_t)row0 * p.y_ld,
                                    rows_here > 0 ? (unsigned)((((int64_t)rows_here - 1) * p.y_ld + p.N) * 2) : 0u);
  const unsigned ystep = (unsigned)(p.y_ld * 16);   // 8 rows down, in bytes
  unsigned yvo = (unsigned)((lane >> 3) * p.y_ld * 2 + (lane & 7) * 16);
  auto store_full = [&](int j, int n0) {   // 8 rows x 128 B: one of the eight stores of a finished chunk PAIR
    ir_wave_lds_fence();                   // rows staged by other lanes (ir_common.h)
    const int r = 8 * j + (lane >> 3);
    const u32x4 v = *(const u32x4_alias*)(tb + r * TPITCH + (lane & 7) * 16);
    buffer_store16_async(yrw, v, yvo + (unsigned)j * ystep, (unsigned)n0 * 2);
  };
  auto store_half = [&](int j, int n0) {   // 16 rows x 64 B (left half of the tile): a range's odd last chunk
    ir_wave_lds_fence();
    const int r = 16 * j + (lane >> 2);
    const u32x4 v = *(const u32x4_alias*)(tb + r * TPITCH + (lane & 3) * 16);
    buffer_store16_async(yrw, v, (unsigned)(r * p.y_ld * 2 + (lane & 3) * 16), (unsigned)n0 * 2);
  };

  // statistics of one finished pair = one head of this wave's 64 rows (see the end of the loop body)
  const bool st_on = p.st_ws != nullptr && row0 < p.M;   // M is whole 64-row blocks then: a wave past M has no block (and no slot in st_ws)
  auto pair_stats = [&](int n0) {
    if (n0 < p.st_col0 || n0 >= p.st_col0 + p.st_cols) return;   // wave-uniform
    ir_wave_lds_fence();
    ir_lds_block_stats<T>(tb, TPITCH, p.st_ws + ((int64_t)(row0 / kStatsRows) * (p.st_cols >> 6) + ((n0 - p.st_col0) >> 6)) * 128);
    ir_wave_lds_fence();
  };

  // Iteration i: start the transfer of chunk i+1, put chunk i-1's results into its half of the staging tile and - when
  // that completes a pair (i even) - send the pair on its way in eight stores issued BETWEEN the MFMA groups of chunk i
  // (a write path running at HBM speed back-pressures the issuing wave: eight stores in a row stall it and its lockstep
  // partner workgroup while the matrix pipe idles); compute chunk i; then wait ONLY for chunk i+1: vector memory
  // operations retire in issue order and the stores were issued after it, so they may stay in flight across the barrier.
  f32x16 accA, accB;
  for (int i = 0; i < ncl; ++i) {
    const int c = c_begin + i, cur = i & 1;
    if (i + 1 < ncl) issue_chunk(c + 1, cur ^ 1);   // its slot was last read in iteration i-1
    if (i > 0) {
      stage_block(accA, 0, (c - 1) * NCH, (i - 1) & 1);
      stage_block(accB, 32, (c - 1) * NCH, (i - 1) & 1);
    }
    const bool pair_done = i > 0 && (i & 1) == 0;   // chunks i-2, i-1 are both in the tile
    const unsigned char* Wb = smem + cur * CHUNK_BYTES;
    if (BIAS && c * NCH >= p.scale_cols) {
      // the bias is the accumulators' starting value (a lane's registers walk along n): no add, no bias registers live while
      // a finished chunk is staged - with them the K = 320 kernel reloaded a resident X fragment from scratch in every
      // iteration since round 2, and a scratch reload waits for vmcnt(0), i.e. for the Y stores in flight
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const v4 bv = *(const v4*)(sbias + (c - c_begin) * NCH + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) accA[4 * g + i] = accB[4 * g + i] = (float)bv[i];
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    }
    if (BIAS) __builtin_amdgcn_sched_barrier(0);   // the bias reads retire before the W fragment reads start (registers)
    // W fragments double-buffered by 64-k sub-tile: the four reads of sub-tile s+1 are in flight while the
    // eight MFMAs of sub-tile s issue
    v8 wf[2][4];
    asm volatile("" : "+v"(wread0));
    int wread[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wread[ks] = wread0 ^ (ks << 5);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wf[0][ks] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + wread[ks]);
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
        accA = Tr::mfma(wf[s & 1][ks], xA[4 * s + ks], accA);
        accB = Tr::mfma(wf[s & 1][ks], xB[4 * s + ks], accB);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (pair_done) {   // the eight parts spread over the KS groups
#pragma unroll
        for (int j = (8 * s) / KS; j < (8 * (s + 1)) / KS; ++j) store_full(j, (c - 2) * NCH);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // token statistics of the pair that has just left (round 4, ir_colstats.h): its 64 x 64 block is still in the staging
    // tile, and the W-fragment registers are free until the next chunk - the statistics cost neither a register across the
    // loop nor a byte of memory traffic (the first form re-read the wave's stores after the loop: 84 MB back from the
    // Infinity Cache on the 131072-row projection, +17 us).  Whole heads only: pairs start on multiples of 64 columns.
    if (pair_done && st_on) pair_stats((c - 2) * NCH);
    // the statistics' four stores were issued after the pair's eight: the count below only gets stricter with them
    if (pair_done) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if ((ncl & 1) == 0) {   // the last chunk completes a pair (its partner went into the tile in the last iteration)
    stage_block(accA, 0, (c_end - 1) * NCH, 1);
    stage_block(accB, 32, (c_end - 1) * NCH, 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) store_full(j, (c_end - 2) * NCH);
    if (st_on) pair_stats((c_end - 2) * NCH);
  } else {                // odd number of chunks in this range: the last one leaves alone, in half lines
    stage_block(accA, 0, (c_end - 1) * NCH, 0);
    stage_block(accB, 32, (c_end - 1) * NCH, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_half(j, (c_end - 1) * NCH);
  }
}

// K = 640 (the 32x32-token layer class): 64 rows of X no longer fit one wave's registers, so the contraction
// is split across TWO waves of the workgroup.  8 waves: wave w owns row block (w & 3) and K half (w >> 2), keeps
// its 64 x 320 slice of X in registers exactly like the kernel above and streams the same W chunks (32 columns x
// all of K, one LDS image for all eight waves).  After a chunk the upper-half waves leave their fp32 partial
// tile in LDS (lane-linear, 8 KiB per row block, two chunk parities); the lower-half waves pick it up after the
// chunk barrier, add, and reuse that very LDS region as the staging tile of their transposed stores, which
// again ride between the MFMAs of the next chunk.  LDS: 2 x 40 KiB of W + 64 KiB of partials (+ bias).
template <typename T, int KSH>               // K = 128 * KSH
__global__ void __launch_bounds__(512, 2) linear_ksplit_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int KS2 = 2 * KSH;
  constexpr int CHUNK_BYTES = KS2 * SUB_BYTES;
  constexpr int R_OFF = 2 * CHUNK_BYTES;     // partial tiles: [parity][row block] x 8 KiB
  constexpr int TPITCH = 80;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  unsigned char* const smem = dsm;
  __shared__ __attribute__((aligned(16))) T sbias[kLinearMaxBiasN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rb = wid & 3, kh = wid >> 2;
  const int hi = lane >> 5, lq = lane & 31;

  const int mb = blockIdx.x / p.nsplit, sp = blockIdx.x - mb * p.nsplit;
  const int nchunks = p.N / NCH;
  const int unit = p.st_ws != nullptr ? 2 : 1, nu = nchunks / unit;   // with the statistics tail: ranges of whole heads (64 columns)
  const int c_begin = unit * (int)(((long)nu * sp) / p.nsplit), c_end = unit * (int)(((long)nu * (sp + 1)) / p.nsplit);
  if (c_begin >= c_end) return;

  // ---- this wave's half of the X rows: resident for the whole kernel -------------------------------
  const int rowA = mb * 256 + rb * 64 + lq, rowB = rowA + 32;
  v8 xA[4 * KSH], xB[4 * KSH];
  {
    const int ra = rowA < p.M ? rowA : p.M - 1, rb2 = rowB < p.M ? rowB : p.M - 1;
    if (p.x_f32) {   // fp32 activations: cast on the way into the resident fragments (see linear_skinny_kernel)
      const float* base = (const float*)p.x + kh * (64 * KSH) + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4 * KSH; ++ks)
        xA[ks] = __builtin_convertvector(*(const f32x8*)(base + (int64_t)ra * p.x_ld + ks * 16), v8);
#pragma unroll
      for (int ks = 0; ks < 4 * KSH; ++ks) asm volatile("" : "+v"(xA[ks]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4 * KSH; ++ks)
        xB[ks] = __builtin_convertvector(*(const f32x8*)(base + (int64_t)rb2 * p.x_ld + ks * 16), v8);
    } else {
      const T* base = (const T*)p.x + kh * (64 * KSH) + hi * 8;
#pragma unroll
      for (int ks = 0; ks < 4 * KSH; ++ks) {
        xA[ks] = *(const v8*)(base + (int64_t)ra * p.x_ld + ks * 16);
        xB[ks] = *(const v8*)(base + (int64_t)rb2 * p.x_ld + ks * 16);
      }
    }
  }

  // ---- W chunk stream: sub-tile s = 2 i + (tid >> 8), 16 B per thread, lane-linear LDS image -----------
  const int t2 = tid & 255, half = wid >> 2;   // wave-uniform: the LDS destination of a transfer lives in M0
  const int wrow = t2 >> 3, wslot = t2 & 7;
  const i32x4 wrw = make_rsrc_words(p.w, (unsigned)(((int64_t)(p.N - 1) * p.w_ld + 128 * KSH) * 2));
  const unsigned wvo = (unsigned)(wrow * p.w_ld * 2 + ((wslot ^ ((wrow >> 1) & 7)) * 16));
  auto issue_chunk = [&](int c, int slot) {
    const unsigned off = wvo + (unsigned)((int64_t)c * NCH * p.w_ld * 2);
#pragma unroll
    for (int i = 0; i < KSH; ++i) {
      const int s = 2 * i + half;
      buffer_load_lds16_async(wrw, smem + slot * CHUNK_BYTES + s * SUB_BYTES + (wid & 3) * 1024, off + s * 128);
    }
  };
  int wread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wread[ks] = lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);

  if (p.bias != nullptr)
    for (int i = tid; i < (c_end - c_begin) * NCH; i += 512) sbias[i] = ((const T*)p.bias)[c_begin * NCH + i];
  const int ncl = c_end - c_begin;
  issue_chunk(c_begin, 0);
#pragma unroll
  for (int ks = 0; ks < 4 * KSH; ++ks) asm volatile("" ::"v"(xA[ks]), "v"(xB[ks]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto rtile = [&](int par) { return smem + R_OFF + (par * 4 + rb) * 8192; };
  // upper half: leave the partial tile (2 x 16 fp32 per lane) lane-linear in LDS
  auto put_partial = [&](const f32x16& a, const f32x16& b, int par) {
    unsigned char* r = rtile(par) + lane * 16;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 fa, fb;
#pragma unroll
      for (int i = 0; i < 4; ++i) { fa[i] = a[4 * g + i]; fb[i] = b[4 * g + i]; }
      *(IR_LDS f32x4*)(IR_LDS unsigned char*)(r + g * 1024) = fa;
      *(IR_LDS f32x4*)(IR_LDS unsigned char*)(r + (4 + g) * 1024) = fb;
    }
  };
  // lower half: add the partner's partial, add the bias, round, and stage the 64 x 32 tile for the transposed stores
  // in the same LDS region (this wave's reads of it are complete before its writes: LDS operations of a wave execute in order)
  auto finish_tile = [&](f32x16& a, f32x16& b, int par, int n0) {
    unsigned char* r = rtile(par);
    auto rd = [&](int q) -> f32x4 { return *(const IR_LDS f32x4_alias*)(IR_LDS unsigned char*)(r + lane * 16 + q * 1024); };   // may_alias: these reads must stay ahead of the 16-bit writes below
    auto bias4 = [&](int g) {
      f32x4 f = {0.f, 0.f, 0.f, 0.f};
      if (p.bias != nullptr) {
        const v4 bv = *(const v4*)(sbias + (n0 - c_begin * NCH) + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = (float)bv[i];
      }
      return f;
    };
    // the staged rows of block A (bytes 0 .. 2559 of the tile) cover the partial quads 0-2 of A; those of block B
    // (2560 .. 5119) cover A's quad 3 and B's quad 0 (4096 ..): read what a write is about to cover first
    const float cs = n0 < p.scale_cols ? p.col_scale : 1.0f;   // leading columns scaled in fp32 before the one rounding
    f32x4 pa[4], pb0;
#pragma unroll
    for (int g = 0; g < 4; ++g) pa[g] = rd(g);
    pb0 = rd(4);
    ir_wave_lds_fence();   // the 16-bit writes below land on partial quads other lanes have just read
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bz = bias4(g);
      f32x4 fa;
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = (a[4 * g + i] + pa[g][i]) * cs + bz[i];
      *(v4*)(r + lq * TPITCH + (8 * g + 4 * hi) * 2) = __builtin_convertvector(fa, v4);
    }
    f32x4 pb[4];
    pb[0] = pb0;
#pragma unroll
    for (int g = 1; g < 4; ++g) pb[g] = rd(4 + g);
    ir_wave_lds_fence();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bz = bias4(g);
      f32x4 fb;
#pragma unroll
      for (int i = 0; i < 4; ++i) fb[i] = (b[4 * g + i] + pb[g][i]) * cs + bz[i];
      *(v4*)(r + (32 + lq) * TPITCH + (8 * g + 4 * hi) * 2) = __builtin_convertvector(fb, v4);
    }
  };
  const int row0 = mb * 256 + rb * 64;
  auto store_part = [&](int j, int par, int n0) {   // 16 rows x 64 B of the staged tile
    ir_wave_lds_fence();
    const int rr = 16 * j + (lane >> 2);
    const u32x4 v = *(const u32x4_alias*)(rtile(par) + rr * TPITCH + (lane & 3) * 16);
    const int row = row0 + rr;
    T* yp = (T*)p.y + (int64_t)(row < p.M ? row : p.M - 1) * p.y_ld + n0 + (lane & 3) * 8;
    *(u32x4*)yp = v;
  };

  f32x16 accA, accB;
  for (int i = 0; i < ncl; ++i) {
    const int c = c_begin + i, cur = i & 1;
    if (i + 1 < ncl) issue_chunk(c + 1, cur ^ 1);
    const bool fin = (kh == 0) && (i > 0);   // chunk i-1 is finished by the lower-half wave while it computes chunk i
    if (fin) finish_tile(accA, accB, (i - 1) & 1, (c - 1) * NCH);
    const unsigned char* Wb = smem + cur * CHUNK_BYTES + kh * KSH * SUB_BYTES;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    // W fragments one half sub-tile ahead: the two reads of the next group are in flight under four MFMAs (a full
    // double buffer costs 16 registers this kernel does not have)
    v8 wa[2], wb[2];
    wa[0] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + wread[0]);
    wa[1] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + wread[1]);
#pragma unroll
    for (int s = 0; s < KSH; ++s) {
      wb[0] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + s * SUB_BYTES + wread[2]);
      wb[1] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + s * SUB_BYTES + wread[3]);
      __builtin_amdgcn_sched_barrier(0);
      accA = Tr::mfma(wa[0], xA[4 * s + 0], accA);
      accB = Tr::mfma(wa[0], xB[4 * s + 0], accB);
      accA = Tr::mfma(wa[1], xA[4 * s + 1], accA);
      accB = Tr::mfma(wa[1], xB[4 * s + 1], accB);
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < KSH) {
        wa[0] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + (s + 1) * SUB_BYTES + wread[0]);
        wa[1] = *(const IR_LDS v8*)(IR_LDS unsigned char*)(Wb + (s + 1) * SUB_BYTES + wread[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      accA = Tr::mfma(wb[0], xA[4 * s + 2], accA);
      accB = Tr::mfma(wb[0], xB[4 * s + 2], accB);
      accA = Tr::mfma(wb[1], xA[4 * s + 3], accA);
      accB = Tr::mfma(wb[1], xB[4 * s + 3], accB);
      __builtin_amdgcn_sched_barrier(0);
      if (fin) {
        if (s < 4) store_part(s, (i - 1) & 1, (c - 1) * NCH);
        if (s == KSH - 1) {
#pragma unroll
          for (int j = KSH; j < 4; ++j) store_part(j, (i - 1) & 1, (c - 1) * NCH);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kh == 1) put_partial(accA, accB, i & 1);
    // the next chunk has landed: the four stores of a finishing wave were issued after its transfers and may stay in flight
    if (fin) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (kh == 0) {
    const int n0 = (c_end - 1) * NCH, par = (ncl - 1) & 1;
    finish_tile(accA, accB, par, n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_part(j, par, n0);
  }
  // ---- tail (round 4): token statistics of the V columns this wave has just written (ir_colstats.h) -----------------
  if (p.st_ws != nullptr && kh == 0 && row0 < p.M) {   // waves past M (M % 256 != 0) have no block and no slot in st_ws
    int head_lo, head_hi;
    ir_stats_heads(p.st_col0, p.st_cols, c_begin * NCH, c_end * NCH, head_lo, head_hi);   // the launcher aligns column ranges to 64 then
    if (head_lo < head_hi) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ir_wave_col_stats<T, 5>((const T*)p.y, p.y_ld, row0, head_lo, head_hi, p.st_ws, p.st_col0, p.st_cols);
    }
  }
}

template <typename T, int KS, bool BIAS, int NW>
hipError_t launch_skinny2(const LinearKParams& p, int grid, hipStream_t s) {
  // two W chunks + NW staging tiles + the bias of at most N columns: 78 KiB at K = 320 with 4 waves (two workgroups per CU
  // fit while the bias stays under ~2 KiB, i.e. N <= 960; wider biased outputs run one workgroup per CU), 114 KiB with 8
  const size_t fixed = (size_t)2 * KS * SUB_BYTES + (size_t)NW * 64 * kSkinnyTPitch;
  const size_t dyn = fixed + (BIAS ? (size_t)p.N * sizeof(T) : 0);
  static IrOncePerDevice once;   // per instantiation (ir_common.h)
  const hipError_t ea = ir_opt_in_dynamic_lds(once, (const void*)linear_skinny_kernel<T, KS, BIAS, NW>,
                                              fixed + (BIAS ? kLinearMaxBiasN * sizeof(T) : 0));
  if (ea != hipSuccess) return ea;
  hipLaunchKernelGGL((linear_skinny_kernel<T, KS, BIAS, NW>), dim3((unsigned)grid), dim3(NW * 64), dyn, s, p);
  return hipGetLastError();
}

template <typename T, int KS>
hipError_t launch_skinny(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  // 8-wave (512-row, one per CU) workgroups measured SLOWER than two 4-wave ones per CU on every K = 320 shape of the step
  // (135 vs 119-125 us on 131072 x 960, 52 vs 48 on x 320, 47 vs 44 on 32768 x 960; profiles/r2_kernel_experiments.txt 7):
  // development builds only (-DIR_ABLATIONS -DLIN_NW8_MIN_M=<rows>)
  constexpr bool w8 = false;
  const int rows = w8 ? 512 : 256, slots = w8 ? 256 : 512;
  const int mblocks = (p.M + rows - 1) / rows;
  const int nchunks = p.N / NCH;
  int nsplit = (slots + mblocks - 1) / mblocks;
  const int nunits = (nchunks & 1) ? nchunks : nchunks / 2;   // ranges start on even chunks when they can (whole-line stores)
  if (nsplit > nunits) nsplit = nunits;
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
  const int grid = mblocks * nsplit;
  return p.bias != nullptr ? launch_skinny2<T, KS, true, 4>(p, grid, s) : launch_skinny2<T, KS, false, 4>(p, grid, s);
}

template <typename T>
hipError_t launch(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  const int mblocks = (p.M + 255) / 256;
  const int nchunks = p.N / NCH;
  if (p.K == 640) {   // contraction split over two waves, one 8-wave workgroup per CU
    int nsplit = (256 + mblocks - 1) / mblocks;
    if (nsplit > nchunks / (p.st_ws != nullptr ? 2 : 1)) nsplit = nchunks / (p.st_ws != nullptr ? 2 : 1);
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
    constexpr int KSH = 5;
    const size_t dyn = (size_t)2 * (2 * KSH) * SUB_BYTES + 2 * 4 * 8192;
    static IrOncePerDevice once;   // per instantiation (ir_common.h)
    const hipError_t ea = ir_opt_in_dynamic_lds(once, (const void*)linear_ksplit_kernel<T, KSH>, dyn);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((linear_ksplit_kernel<T, KSH>), dim3((unsigned)(mblocks * nsplit)), dim3(512), dyn, s, p);
    return hipGetLastError();
  }
  switch (p.K / 64) {
    case 1: return launch_skinny<T, 1>(p, s);
    case 2: return launch_skinny<T, 2>(p, s);
    case 3: return launch_skinny<T, 3>(p, s);
    case 4: return launch_skinny<T, 4>(p, s);
    case 5: return launch_skinny<T, 5>(p, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t ir_launch_linear_skinny(const LinearKParams& p, int dtype, hipStream_t s) {
  return dtype == 1 ? launch<__bf16>(p, s) : launch<_Float16>(p, s);
}
