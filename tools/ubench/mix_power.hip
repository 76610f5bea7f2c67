// Microbenchmark (gfx950): the attention kernel's per-tile instruction mix (32 8-pass MFMAs or 64 4-pass MFMAs,
// 64 v_exp_f32, 64 packed and 128 single fp32 VALU per wave, all waves in the same phase with one barrier per
// tile) on pseudo-random MFMA operands, run long enough for the power controller to settle: does the MFMA form
// change the time under the board power cap?
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int FORM, int VMODE>   // FORM 0: 32x32x16, 1: 16x16x32;  VMODE 0: no vector work, 1: the softmax mix
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, int zero) {
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
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
  const float c1 = 0.999f, c2 = 0.001f;
  f32x16 acc[4];
  f32x4 acc4[8];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc4[j][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (FORM == 0) {
#pragma unroll
      for (int m = 0; m < 32; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m & 7], b[(m * 3) & 7], acc[m & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int m = 0; m < 64; ++m) acc4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[m & 7], b[(m * 3) & 7], acc4[m & 7], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (VMODE) {
#pragma unroll
      for (int j = 0; j < 64; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 31]));
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const int q = (2 * j) & 31;
        f32x2 t = {v[q], v[q + 1]};
        const f32x2 k1 = {c1, c1}, k2 = {c2, c2};
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(k1), "v"(k2));
        v[q] = t[0]; v[q + 1] = t[1];
      }
#pragma unroll
      for (int j = 0; j < 128; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j * 7) & 31]) : "v"(c1), "v"(c2));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += acc[j][i];
  for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) sum += acc4[j][i];
  for (int i = 0; i < 32; ++i) sum += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int FORM, int VMODE>
void run(const char* name, int zero) {
  float* out;
  const int iters = 20000;
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<FORM, VMODE>), dim3(256), dim3(512), 0, 0, out, iters, zero);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<FORM, VMODE>), dim3(256), dim3(512), 0, 0, out, iters, zero);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %s: %8.1f ns per tile pair  (%.0f TFLOP/s)\n", name, zero ? "zeros " : "random", ms * 1e6 / (3.0 * iters),
         3.0 * iters * 32 * 32768.0 * 2048 / (ms * 1e-3) / 1e12);
  (void)hipFree(out);
}

int main() {
  for (int zero = 0; zero <= 1; ++zero) {
    run<0, 1>("32x32x16 + softmax mix, same phase", zero);
    run<1, 1>("16x16x32 + softmax mix, same phase", zero);
    run<0, 0>("32x32x16 only (barrier per 32)", zero);
    run<1, 0>("16x16x32 only (barrier per 64)", zero);
  }
  return 0;
}
