// Microbenchmark (gfx950, round 4): does operand REUSE between consecutive MFMAs change what the matrix pipe costs under the
// board power cap?  Same stream as mfma_power.hip (32x32x16 bf16, two waves per SIMD, 4 accumulators in rotation, pseudo-random
// operands), the order of the operand registers varied:
//   0  A and B both change on every MFMA (mfma_power.hip's pattern)
//   1  A held for 2 consecutive MFMAs, B alternates between 2 registers per A   (the attention kernel: one K fragment, two Q blocks)
//   2  A held for 4 consecutive MFMAs, B walks 4 registers
//   3  A held for 8, B walks 8
//   4  A AND B held for 2 consecutive MFMAs (different accumulators)
//   5  one A, one B for the whole loop (only the accumulators differ): the floor of operand toggling
// Reported: sustained clock and TFLOP/s (higher clock under the cap = fewer joules per MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int PAT>
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, long long* clk, int zero) {
  bf16x8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 8; ++j) {
    u32x4 ua, ub;
    for (int i = 0; i < 4; ++i) {
      s = s * 1664525u + 1013904223u; ua[i] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);
      s = s * 1664525u + 1013904223u; ub[i] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);
    }
    a[j] = __builtin_bit_cast(bf16x8, ua);
    b[j] = __builtin_bit_cast(bf16x8, ub);
  }
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      int ia, ib;
      if (PAT == 0) { ia = m & 7; ib = (m * 3) & 7; }
      else if (PAT == 1) { ia = (m >> 1) & 7; ib = ((m >> 1) * 2 + (m & 1)) & 7; }
      else if (PAT == 2) { ia = (m >> 2) & 7; ib = m & 7; }
      else if (PAT == 3) { ia = (m >> 3) & 7; ib = m & 7; }
      else if (PAT == 4) { ia = (m >> 1) & 7; ib = ((m >> 1) * 3) & 7; }
      else { ia = 0; ib = 0; }
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a[ia]), "v"(b[ib]));
    }
  }
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += acc[j][i];
  const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int PAT>
void run(const char* name, int zero) {
  float* out; long long* clk;
  const int iters = 40000;
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(512), 0, 0, out, iters, clk, zero);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(512), 0, 0, out, iters, clk, zero);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  const double flops = 5.0 * iters * 16 * 32768.0 * 2048;
  printf("%-62s %s: %8.2f ms  %7.1f TFLOP/s  clock %.3f GHz\n", name, zero ? "zeros " : "random", ms / 5, flops / (ms * 1e-3) / 1e12,
         (double)c[0] / ((double)c[1] * 10.0));
  (void)hipFree(out); (void)hipFree(clk);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("0 A and B change on every MFMA", 0);
    run<1>("1 A held for 2 MFMAs, B alternates", 0);
    run<2>("2 A held for 4 MFMAs", 0);
    run<3>("3 A held for 8 MFMAs", 0);
    run<4>("4 A and B held for 2 MFMAs", 0);
    run<5>("5 one A, one B throughout", 0);
  }
  run<0>("0 A and B change on every MFMA", 1);
  return 0;
}
