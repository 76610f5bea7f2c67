// Microbenchmark (gfx950): can ONE wave per SIMD hide VALU work in the shadow of its own MFMAs?
// Each loop iteration issues NM v_mfma_f32_32x32x16_bf16 (two independent accumulator chains) and NV
// independent VALU ops (v_fma_f32 or v_exp_f32 on private registers), interleaved by the program order
// written here (sched_group_barrier pins "1 MFMA, NV/NM VALU").  Grid = 256 CUs x 4 waves (one wave per
// SIMD) or x 8 (two per SIMD).  Prints clocks per iteration (s_memtime) for: MFMA only, VALU only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NM, int NV, bool EXP, bool DO_M, bool DO_V>
__global__ void __launch_bounds__(256, 1) k(float* out, int iters, long long* clk) {
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (DO_M) {
        if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      }
      if (DO_V) {
#pragma unroll
        for (int j = 0; j < NV / NM; ++j) {
          const int r = (m * (NV / NM) + j) & 15;
          if (EXP) v[r] = __builtin_amdgcn_exp2f(v[r]) * 0.5f - 1.0f;   // exp + 1 fma: keeps the value bounded
          else v[r] = __builtin_fmaf(v[r], 0.999f, 0.001f);
        }
      }
      if (DO_M && DO_V) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, (EXP ? 2 : 1) * (NV / NM), 0);   // its VALU group
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *clk = t1 - t0;
}

template <int NM, int NV, bool EXP, bool DO_M, bool DO_V>
double run(int waves_per_simd, int iters, const char* name) {
  float* out; long long* clk;
  const int blocks = 256 * waves_per_simd;   // 256-thread blocks: 4 waves = one per SIMD
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clk, sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NV, EXP, DO_M, DO_V>), dim3(blocks), dim3(256), 0, 0, out, 10, clk);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NV, EXP, DO_M, DO_V>), dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost);
  const double us_per_iter = ms * 1e3 / iters;
  printf("%-34s waves/SIMD %d: %8.3f us/iter  (%6.1f ns)  counter ticks/iter %.1f\n", name, waves_per_simd, us_per_iter,
         us_per_iter * 1e3, (double)c / iters);
  hipFree(out); hipFree(clk);
  return us_per_iter;
}

int main() {
  const int iters = 20000;
  for (int w = 1; w <= 2; ++w) {
    run<8, 0, false, true, false>(w, iters, "8 MFMA");
    run<8, 48, false, false, true>(w, iters, "48 v_fma");
    run<8, 48, false, true, true>(w, iters, "8 MFMA + 48 v_fma (6 per MFMA)");
    run<8, 64, false, true, true>(w, iters, "8 MFMA + 64 v_fma (8 per MFMA)");
    run<8, 16, true, false, true>(w, iters, "16 v_exp (+16 fma)");
    run<8, 16, true, true, true>(w, iters, "8 MFMA + 16 v_exp (+16 fma)");
    run<8, 32, true, true, true>(w, iters, "8 MFMA + 32 v_exp (+32 fma)");
  }
  return 0;
}
