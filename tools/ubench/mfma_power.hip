// Microbenchmark (gfx950): sustained shader clock of MFMA-only streams under the board power cap, as a proxy for
// energy per flop of the instruction form.  Two waves per SIMD, pseudo-random bf16 operands (8 A and 8 B
// fragments rotated so that operand buses toggle), accumulators in VGPRs (builtin) or AGPRs (inline asm).
// Prints ns per iteration, shader ticks (s_memtime) and 100 MHz ticks (s_memrealtime) of wave 0 -> clock.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int FORM>   // 0: 32x32x16 VGPR acc, 1: 32x32x16 AGPR acc, 2: 16x16x32 VGPR acc, 3: 16x16x32 AGPR acc
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, long long* clk, int zero) {
  bf16x8 a[8], b[8];
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 8; ++j) {
    u32x4 ua, ub;
    for (int i = 0; i < 4; ++i) {
      s = s * 1664525u + 1013904223u; ua[i] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);   // +-[0.5, 1) mantissas random
      s = s * 1664525u + 1013904223u; ub[i] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);
    }
    a[j] = __builtin_bit_cast(bf16x8, ua);
    b[j] = __builtin_bit_cast(bf16x8, ub);
  }
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  if (FORM < 2) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (FORM == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a[m & 7]), "v"(b[(m * 3) & 7]));
        else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m * 3) & 7], acc[m & 3], 0, 0, 0);
      }
    }
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += acc[j][i];
  } else {
    f32x4 acc[8];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 32; ++m) {
        if (FORM == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(a[m & 7]), "v"(b[(m * 3) & 7]));
        else acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 7], b[(m * 3) & 7], acc[m & 7], 0, 0, 0);
      }
    }
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) sum += acc[j][i];
  }
  const long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

template <int FORM>
void run(const char* name, int zero) {
  float* out; long long* clk;
  const int iters = 40000;   // ~25 ms per launch: long enough for the power controller to settle
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<FORM>), dim3(256), dim3(512), 0, 0, out, iters, clk, zero);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<FORM>), dim3(256), dim3(512), 0, 0, out, iters, clk, zero);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  const double flops = 5.0 * iters * 16 * 32768.0 * 2048;   // 2048 waves
  printf("%-34s %s: %8.2f ms  %7.1f TFLOP/s  clock %.3f GHz (wave 0: %.1f ticks per 16 8-pass MFMAs)\n", name, zero ? "zeros " : "random", ms / 5,
         flops / (ms * 1e-3) / 1e12, (double)c[0] / ((double)c[1] * 10.0), (double)c[0] / iters);
  (void)hipFree(out); (void)hipFree(clk);
}

int main() {
  for (int zero = 0; zero <= 1; ++zero) {
    run<0>("32x32x16, accumulators in VGPRs", zero);
    run<1>("32x32x16, accumulators in AGPRs", zero);
    run<2>("16x16x32, accumulators in VGPRs", zero);
    run<3>("16x16x32, accumulators in AGPRs", zero);
  }
  return 0;
}
