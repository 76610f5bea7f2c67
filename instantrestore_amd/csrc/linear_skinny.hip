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
#include "ir_kernels.h"

namespace {

constexpr int NCH = 32;                     // columns of Y (rows of W) per chunk
constexpr int SUB_BYTES = NCH * 128;        // one 64-k sub-tile of a chunk: 32 rows x 128 B

template <typename T, int KS>               // K = 64 * KS
__global__ void __launch_bounds__(256, 2) linear_skinny_kernel(const LinearKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  using v4 = typename Tr::v4;
  constexpr int CHUNK_BYTES = KS * SUB_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * CHUNK_BYTES];   // two W chunks: one being read, one in flight
  constexpr int TPITCH = 80;                 // bytes per row of the per-wave output staging tile (64 B + pad)
  __shared__ __attribute__((aligned(16))) unsigned char tbuf[4 * 64 * TPITCH];
  __shared__ __attribute__((aligned(16))) T sbias[kLinearMaxBiasN];   // a global bias load inside the loop would make
                                                                      // hipcc wait on the in-flight W chunk as well

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, lq = lane & 31;

  // ---- work decode: (row block of 256, range of column chunks) -------------------------------------
  const int mb = blockIdx.x / p.nsplit, sp = blockIdx.x - mb * p.nsplit;
  const int nchunks = p.N / NCH;
  const int c_begin = (int)(((long)nchunks * sp) / p.nsplit), c_end = (int)(((long)nchunks * (sp + 1)) / p.nsplit);
  if (c_begin >= c_end) return;

  // ---- X fragments of both 32-row blocks: resident for the whole kernel ------------------------------
  const int rowA = mb * 256 + wid * 64 + lq, rowB = rowA + 32;
  v8 xA[4 * KS], xB[4 * KS];
  {
    const int ra = rowA < p.M ? rowA : p.M - 1, rb = rowB < p.M ? rowB : p.M - 1;
    const T* base = (const T*)p.x + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4 * KS; ++ks) {
      xA[ks] = *(const v8*)(base + (int64_t)ra * p.x_ld + ks * 16);
      xB[ks] = *(const v8*)(base + (int64_t)rb * p.x_ld + ks * 16);
    }
  }

  // ---- W chunk stream: 16 B per thread per sub-tile, lane-linear LDS image, swizzle on the source ------
  const int wrow = tid >> 3, wslot = tid & 7;
  const i32x4 wrw = make_rsrc_words(p.w, (unsigned)(((int64_t)(p.N - 1) * p.w_ld + 64 * KS) * 2));
  const unsigned wvo = (unsigned)(wrow * p.w_ld * 2 + ((wslot ^ ((wrow >> 1) & 7)) * 16));
  auto issue_chunk = [&](int c, int slot) {
    const unsigned off = wvo + (unsigned)((int64_t)c * NCH * p.w_ld * 2);
#pragma unroll
    for (int s = 0; s < KS; ++s)
      buffer_load_lds16_async(wrw, smem + slot * CHUNK_BYTES + s * SUB_BYTES + wid * 1024, off + s * 128);
  };
  int wread[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) wread[ks] = lq * 128 + (((2 * ks + hi) ^ ((lq >> 1) & 7)) << 4);

  if (p.bias != nullptr)
    for (int i = tid; i < (c_end - c_begin) * NCH; i += 256) sbias[i] = ((const T*)p.bias)[c_begin * NCH + i];
  const int ncl = c_end - c_begin;
  issue_chunk(c_begin, 0);
#pragma unroll
  for (int ks = 0; ks < 4 * KS; ++ks) asm volatile("" ::"v"(xA[ks]), "v"(xB[ks]));  // X loads retire before the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // A finished chunk (64 rows x 32 columns per wave) leaves through a wave-private LDS tile: in the MFMA
  // layout a lane owns 4-column groups of ONE row, so direct stores touch 32 rows with 16-32 B each and
  // the write path drowns in partial-line requests; after the transpose four neighbouring lanes write
  // the 64 contiguous bytes of a row (16 B each, 16 rows per store instruction).
  unsigned char* tb = tbuf + wid * (64 * TPITCH);
  auto stage_block = [&](const f32x16& acc, int rbase, int n0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 f;
#pragma unroll
      for (int i = 0; i < 4; ++i) f[i] = acc[4 * g + i];
      if (p.bias != nullptr) {
        const v4 bv = *(const v4*)(sbias + (n0 - c_begin * NCH) + 8 * g + 4 * hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] += (float)bv[i];
      }
      *(v4*)(tb + (rbase + lq) * TPITCH + (8 * g + 4 * hi) * 2) = __builtin_convertvector(f, v4);
    }
  };
  // rows past M were loaded as copies of row M-1 and therefore hold row M-1's exact result: they are
  // stored there too (same bytes), which keeps the number of stores per iteration constant - the
  // s_waitcnt arithmetic of the main loop counts them
  const int row0 = mb * 256 + wid * 64;
  auto store_part = [&](int j, int n0) {   // 16 rows x 64 B: one of the four stores of a finished chunk
    const int r = 16 * j + (lane >> 2);
    const u32x4 v = *(const u32x4*)(tb + r * TPITCH + (lane & 3) * 16);
    const int row = row0 + r;
    T* yp = (T*)p.y + (int64_t)(row < p.M ? row : p.M - 1) * p.y_ld + n0 + (lane & 3) * 8;
    *(u32x4*)yp = v;
  };
  auto store_chunk = [&](const f32x16& a, const f32x16& b, int n0) {
    stage_block(a, 0, n0);
    stage_block(b, 32, n0);
#pragma unroll
    for (int j = 0; j < 4; ++j) store_part(j, n0);
  };

  // Iteration i: start the transfer of chunk i+1, send chunk i-1's results on their way, compute chunk i,
  // then wait ONLY for chunk i+1: vector memory operations retire in issue order and the four stores
  // were issued after it, so they may stay in flight across the barrier.
  f32x16 accA, accB;
  for (int i = 0; i < ncl; ++i) {
    const int c = c_begin + i, cur = i & 1;
    if (i + 1 < ncl) issue_chunk(c + 1, cur ^ 1);   // its slot was last read in iteration i-1
    // chunk i-1 leaves in four 16-row stores that are issued BETWEEN the MFMA groups of chunk i: a write
    // path running at HBM speed back-pressures the issuing wave, and four stores in a row stall it (and
    // its lockstep partner workgroup) while the matrix pipe idles
    if (i > 0) {
      stage_block(accA, 0, (c - 1) * NCH);
      stage_block(accB, 32, (c - 1) * NCH);
    }
    const unsigned char* Wb = smem + cur * CHUNK_BYTES;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    // W fragments double-buffered by 64-k sub-tile: the four reads of sub-tile s+1 are in flight while the
    // eight MFMAs of sub-tile s issue
    v8 wf[2][4];
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
      if (i > 0) {   // KS < 4: the remaining parts go out after the last group
        if (s < 4) store_part(s, (c - 1) * NCH);
        if (s == KS - 1) {
#pragma unroll
          for (int j = KS; j < 4; ++j) store_part(j, (c - 1) * NCH);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (i > 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  store_chunk(accA, accB, (c_end - 1) * NCH);
}

template <typename T>
hipError_t launch(const LinearKParams& p0, hipStream_t s) {
  LinearKParams p = p0;
  const int mblocks = (p.M + 255) / 256;
  const int nchunks = p.N / NCH;
  int nsplit = (512 + mblocks - 1) / mblocks;       // two workgroups per CU
  if (nsplit > nchunks) nsplit = nchunks;
  if (nsplit < 1) nsplit = 1;
  p.nsplit = nsplit;
  const dim3 g((unsigned)(mblocks * nsplit)), t(256);
  switch (p.K / 64) {
    case 1: hipLaunchKernelGGL((linear_skinny_kernel<T, 1>), g, t, 0, s, p); break;
    case 2: hipLaunchKernelGGL((linear_skinny_kernel<T, 2>), g, t, 0, s, p); break;
    case 3: hipLaunchKernelGGL((linear_skinny_kernel<T, 3>), g, t, 0, s, p); break;
    case 4: hipLaunchKernelGGL((linear_skinny_kernel<T, 4>), g, t, 0, s, p); break;
    case 5: hipLaunchKernelGGL((linear_skinny_kernel<T, 5>), g, t, 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

hipError_t ir_launch_linear_skinny(const LinearKParams& p, int dtype, hipStream_t s) {
  return dtype == 1 ? launch<__bf16>(p, s) : launch<_Float16>(p, s);
}
