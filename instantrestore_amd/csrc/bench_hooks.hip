// bench_hooks.hip - measurement kernels behind the C ABI's benchmark hooks (not on the product path).
//
// ir_bench_mfma_stream: an MFMA-only stream of the attention kernels' instruction (v_mfma_f32_32x32x16_{bf16,f16}), two
// waves per SIMD on every CU, sixteen MFMAs per iteration over four accumulators, operands either pseudo-random or all zero.
// bench.py runs it for a few tenths of a second so that the board's power controller settles, and reports what the matrix
// pipe SUSTAINS on this box under its 1400 W cap: the 2.5 PFLOP/s the contract's `roofline.frac` divides by assumes 2.4 GHz,
// which random operands never see (profiles/r1_ubench_mfma_power.txt, r4_ubench_mfma_reuse.txt: 1.92-1.99 PFLOP/s).
#include <type_traits>

#include "ir_common.h"
#include "ir_kernels.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(512, 2) mfma_stream_kernel(float* out, int iters, int zero) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  v8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  // 16-bit patterns with random sign and mantissa and an exponent that keeps |x| in [0.5, 1): finite products, toggling operands
  const unsigned keep = std::is_same<T, __bf16>::value ? 0x807f807fu : 0x83ff83ffu;
  const unsigned expo = std::is_same<T, __bf16>::value ? 0x3f003f00u : 0x38003800u;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    u32x4 ua, ub;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s = s * 1664525u + 1013904223u; ua[i] = zero ? 0u : ((s & keep) | expo);
      s = s * 1664525u + 1013904223u; ub[i] = zero ? 0u : ((s & keep) | expo);
    }
    a[j] = __builtin_bit_cast(v8, ua);
    b[j] = __builtin_bit_cast(v8, ub);
  }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m & 3] = Tr::mfma(a[m & 7], b[(m * 3) & 7], acc[m & 3]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[j][i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;   // keeps the stream alive; one float per thread
}

}  // namespace

// one launch: `blocks` workgroups of 8 waves, `iters` x 16 MFMAs per wave; out: blocks * 512 floats
hipError_t ir_launch_bench_mfma_stream(int dtype, int zero, int iters, int blocks, float* out, hipStream_t s) {
  if (dtype == 1) hipLaunchKernelGGL(mfma_stream_kernel<__bf16>, dim3(blocks), dim3(512), 0, s, out, iters, zero);
  else hipLaunchKernelGGL(mfma_stream_kernel<_Float16>, dim3(blocks), dim3(512), 0, s, out, iters, zero);
  return hipGetLastError();
}
