// Microbenchmark (gfx950): the schedule the attention kernel would need.  One 8-wave workgroup per CU; waves
// 0-3 and 4-7 (one of each per SIMD) run the same loop { vector phase; s_barrier; matrix phase; s_barrier } one
// phase apart, so on every SIMD one wave is in its matrix phase (NM MFMAs, each followed by `s_nop 1`: see
// overlap_types.hip) while the other is in its vector phase (NE v_exp_f32, NP packed fp32 ops, NS single-issue
// fp32 ops - inline asm, so the mix is exactly what is written).  Reports ns and wave-0 ticks per period.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int NM, int NE, int NP, int NS, int MODE, int GAP>   // MODE 0: ping-pong, 1: same phase (all waves M then V), 2: M only, 3: V only
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, long long* clk) {
  const long long c0 = __builtin_readcyclecounter();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
  const float c1 = 0.999f, c2 = 0.001f;
  const int group = threadIdx.x >> 8;
  auto mphase = [&]() {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      if (GAP) { asm volatile("s_nop 1"); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  auto vphase = [&]() {
#pragma unroll
    for (int j = 0; j < NE; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j & 31]));
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int q = (2 * j) & 31;
      f32x2 t = {v[q], v[q + 1]};
      const f32x2 k1 = {c1, c1}, k2 = {c2, c2};
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(k1), "v"(k2));
      v[q] = t[0]; v[q + 1] = t[1];
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(j * 7) & 31]) : "v"(c1), "v"(c2));
  };
  if (MODE == 0 && group == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one phase behind
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 1) {
      vphase();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      mphase();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 4) {   // one mixed stream per wave: MFMA, (gap), its share of the vector work
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        if (GAP) asm volatile("s_nop 1");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NE / NM; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(v[(m * 2 + j) & 31]));
#pragma unroll
        for (int j = 0; j < NP / NM; ++j) {
          const int q = (2 * (m * 2 + j)) & 31;
          f32x2 t = {v[q], v[q + 1]};
          const f32x2 k1 = {c1, c1}, k2 = {c2, c2};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(k1), "v"(k2));
          v[q] = t[0]; v[q + 1] = t[1];
        }
#pragma unroll
        for (int j = 0; j < NS / NM; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[((m * 9 + j) * 7) & 31]) : "v"(c1), "v"(c2));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 2) {
      mphase();
    } else {
      vphase();
    }
  }
  if (MODE == 0 && group == 0) __builtin_amdgcn_s_barrier();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = __builtin_readcyclecounter() - c0;
}

template <int NM, int NE, int NP, int NS, int MODE, int GAP>
void run(const char* name) {
  float* out; long long* clk;
  const int iters = 5000;
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NE, NP, NS, MODE, GAP>), dim3(256), dim3(512), 0, 0, out, 10, clk);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NE, NP, NS, MODE, GAP>), dim3(256), dim3(512), 0, 0, out, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  printf("%-64s %8.1f ns per period  %8.1f wave-0 ticks\n", name, ms * 1e6 / iters, (double)c / iters);
  (void)hipFree(out); (void)hipFree(clk);
}

#define SET(NE, NP, NS, label)                                                              \
  run<32, NE, NP, NS, 2, 0>(label ": matrix phase alone (2 waves/SIMD)");                   \
  run<32, NE, NP, NS, 3, 0>(label ": vector phase alone (2 waves/SIMD)");                   \
  run<32, NE, NP, NS, 1, 0>(label ": all waves in the same phase");                         \
  run<32, NE, NP, NS, 0, 0>(label ": ping-pong");                                           \
  run<32, NE, NP, NS, 0, 1>(label ": ping-pong, s_nop 1 after each MFMA");   \
  run<32, NE, NP, NS, 4, 0>(label ": mixed stream, no gap");                                \
  run<32, NE, NP, NS, 4, 1>(label ": mixed stream, s_nop 1 after each MFMA");

int main() {
  // per 64-row x 64-key wave-tile the attention kernel issues 32 MFMAs, 64 v_exp, ~64 packed and ~140 single VALU
  SET(64, 64, 128, "64 exp + 64 pk + 128 single")
  SET(64, 0, 256, "64 exp +  0 pk + 256 single")
  SET(64, 32, 192, "64 exp + 32 pk + 192 single")
  SET(64, 0, 192, "64 exp +  0 pk + 192 single")
  return 0;
}
