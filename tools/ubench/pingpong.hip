// Microbenchmark (gfx950): two wave groups of one 8-wave workgroup (one wave of each group per SIMD)
// alternating a matrix phase (NM MFMAs, two chains) and a vector phase (NE v_exp + NF v_fma) with one
// s_barrier per phase.  Group B starts half a period late, so on every SIMD one wave is in its matrix
// phase while the other is in its vector phase.  Reports time per (M + V) period against the phases alone.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int NM, int NE, int NF, int MODE, bool AG = false, int GAP = 0>   // AG: accumulators in AGPRs (inline asm)
// MODE 0: ping-pong with barriers, 1: same phases, no barriers (free running), 2: M only, 3: V only
__global__ void __launch_bounds__(512, 2) k(float* out, int iters, long long* clk) {
  const long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
  const int group = (threadIdx.x >> 6) >> 2;   // waves 0-3: group A, 4-7: group B
  auto mphase = [&]() {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 3]) : "v"(a), "v"(b));
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      if (GAP == 1) { asm volatile("s_nop 1"); __builtin_amdgcn_sched_barrier(0); }
      if (GAP == 2) { asm volatile("s_nop 3"); __builtin_amdgcn_sched_barrier(0); }
    }
  };
  auto vphase = [&]() {
#pragma unroll
    for (int j = 0; j < NE; ++j) v[j & 31] = __builtin_amdgcn_exp2f(v[j & 31]) * 0.5f - 1.0f;
#pragma unroll
    for (int j = 0; j < NF; ++j) v[(j * 7) & 31] = __builtin_fmaf(v[(j * 7) & 31], 0.999f, 0.001f);
  };
  if (MODE == 0 && group == 1) { vphase(); }   // B starts half a period late: A: M V M V ..., B: V M V M ...
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      if (group == 0) mphase(); else mphase();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      vphase();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
    } else if (MODE == 1) {
      mphase();
      __builtin_amdgcn_sched_barrier(0);
      vphase();
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 2) {
      mphase();
    } else if (MODE == 3) {
      vphase();
    } else if (MODE == 4) {   // one stream, 1 MFMA : (2 NE + NF) / NM VALU, pinned
      mphase();
      vphase();
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, (2 * NE + NF) / NM, 0);
      }
    } else if (MODE == 5 || MODE == 6) {   // free running, the matrix (5) / vector (6) phase at raised priority
      __builtin_amdgcn_s_setprio(MODE == 5 ? 2 : 0);
      mphase();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(MODE == 5 ? 0 : 2);
      vphase();
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 7) {   // free running, group B starts with its vector phase (no barriers)
      if (group == 0) { mphase(); __builtin_amdgcn_sched_barrier(0); vphase(); }
      else { vphase(); __builtin_amdgcn_sched_barrier(0); mphase(); }
      __builtin_amdgcn_sched_barrier(0);
    } else {   // 8: half tiles - M/2 V/2 M/2 V/2, free running
#pragma unroll
      for (int hlf = 0; hlf < 4; ++hlf) {
#pragma unroll
        for (int m = 0; m < NM / 4; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NE / 4; ++j) v[j & 31] = __builtin_amdgcn_exp2f(v[j & 31]) * 0.5f - 1.0f;
#pragma unroll
        for (int j = 0; j < NF / 4; ++j) v[(j * 7) & 31] = __builtin_fmaf(v[(j * 7) & 31], 0.999f, 0.001f);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - c0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int NM, int NE, int NF, int MODE, bool AG = false, int GAP = 0>
void run(const char* name) {
  float* out;
  const int iters = 5000;
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  long long* clk; (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, NE, NF, MODE, AG, GAP>), dim3(256), dim3(512), 0, 0, out, 10, clk);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((k<NM, NE, NF, MODE, AG, GAP>), dim3(256), dim3(512), 0, 0, out, iters, clk);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
  printf("%-46s %8.1f ns per period  %8.1f s_memtime ticks  %7.2f realtime ticks (100 MHz)\n", name, ms * 1e6 / iters, (double)c[0] / iters,
         (double)c[1] / iters);
  (void)hipFree(clk);
  (void)hipFree(out);
}

int main() {
  // one 64-row wave-tile: 32 MFMAs, 64 exp, ~216 other VALU;  one 32-row wave-tile: 16 / 32 / 100
  run<32, 64, 216, 2>("64-row tile: matrix phase alone (2 waves/SIMD)");
  run<32, 64, 216, 3>("64-row tile: vector phase alone (2 waves/SIMD)");
  run<32, 64, 216, 1>("64-row tile: free running, no barriers");
  run<32, 64, 216, 0>("64-row tile: ping-pong, barrier per phase");
  run<32, 64, 216, 4>("64-row tile: one stream, 1 MFMA : 10.75 VALU");
  run<32, 64, 216, 5>("64-row tile: free running, matrix phase prio 2");
  run<32, 64, 216, 6>("64-row tile: free running, vector phase prio 2");
  run<32, 64, 216, 7>("64-row tile: free running, B starts in V");
  run<32, 64, 216, 8>("64-row tile: free running, quarter phases");
  run<32, 64, 216, 2, false, 1>("s_nop 1 after each MFMA: matrix phase alone");
  run<32, 64, 216, 1, false, 1>("s_nop 1: free running");
  run<32, 64, 216, 0, false, 1>("s_nop 1: ping-pong, barrier per phase");
  run<32, 64, 216, 7, false, 1>("s_nop 1: free running, B starts in V");
  run<32, 64, 216, 0, false, 2>("s_nop 3: ping-pong, barrier per phase");
  run<32, 64, 216, 7, false, 2>("s_nop 3: free running, B starts in V");
  run<32, 64, 216, 2, true>("AGPR acc: matrix phase alone");
  run<32, 64, 216, 1, true>("AGPR acc: free running, no barriers");
  run<32, 64, 216, 0, true>("AGPR acc: ping-pong, barrier per phase");
  run<32, 64, 216, 7, true>("AGPR acc: free running, B starts in V");
  run<16, 32, 100, 2>("32-row tile: matrix phase alone");
  run<16, 32, 100, 3>("32-row tile: vector phase alone");
  run<16, 32, 100, 1>("32-row tile: free running, no barriers");
  run<16, 32, 100, 0>("32-row tile: ping-pong, barrier per phase");
  return 0;
}
