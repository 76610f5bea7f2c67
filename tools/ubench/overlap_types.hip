// Microbenchmark (gfx950): which VALU instruction classes make progress on a SIMD while ANOTHER wave of the
// same SIMD keeps the matrix pipe busy?  One 8-wave workgroup per CU: waves 0-3 (one per SIMD) run MFMAs only,
// waves 4-7 run one VALU instruction form only (inline asm, 32 independent registers).  Each group reports its
// own s_memtime ticks; the run is repeated with only one of the groups active.  "both" ~ max(alone) means the
// class overlaps with the matrix pipe, "both" ~ sum means it does not.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// VT: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_add_u32, 4 v_mov_b32, 5 v_cvt_pk_bf16_f32, 6 v_max3_f32, 7 v_pk_add_f32, 8 v_mul_f32 (e32), 9 v_and_b32
template <int VT, int MT>   // MT: 0 = 32x32x16 bf16 (8 passes), 1 = 16x16x32 bf16 (4 passes)
__global__ void __launch_bounds__(512, 2) k(float* out, int iters_m, int iters_v, int active, long long* clk, int swap, int prio) {
  const int group = (threadIdx.x >> 8) ^ swap;   // swap: the VALU waves are the OLDER waves of each SIMD
  if (prio && group == 1) __builtin_amdgcn_s_setprio(3);
  if (prio == 2 && group == 0) __builtin_amdgcn_s_setprio(3);
  float s = 0.f;
  const long long c0 = __builtin_readcyclecounter();
  if (group == 0) {
    if (active & 1) {
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
      if (MT != 1) {
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
          for (int m = 0; m < 16; ++m) {
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            // paced: the wave does not come back for the VALU port until the matrix pipe is (nearly) free again
            if (MT == 2) asm volatile("s_nop 15\n\ts_nop 11" ::: "memory");
            if (MT == 3) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            if (MT == 4) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            if (MT == 5) asm volatile("s_nop 15\n\ts_nop 9" ::: "memory");
            if (MT == 6) asm volatile("s_nop 0" ::: "memory");
            if (MT == 7) asm volatile("s_nop 1" ::: "memory");
            if (MT == 8) asm volatile("s_nop 3" ::: "memory");
          }
        }
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
      } else {
        f32x4 acc[8];
        for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
        for (int it = 0; it < iters_m; ++it) {
#pragma unroll
          for (int m = 0; m < 32; ++m) acc[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m & 7], 0, 0, 0);
        }
        for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i];
      }
    }
  } else if (active & 2) {
    float v[32];
    for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
    const float c1 = 0.999f, c2 = 0.001f;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const int r = j & 31;
        if (VT == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(c1), "v"(c2));
        else if (VT == 1) {
          const int q = (2 * j) & 31;
          f32x2 t = {v[q], v[q + 1]};
          const f32x2 k1 = {c1, c1}, k2 = {c2, c2};
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(k1), "v"(k2));
          v[q] = t[0]; v[q + 1] = t[1];
        } else if (VT == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
        else if (VT == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[r]) : "v"(c1));
        else if (VT == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(v[r]) : "v"(c1));
        else if (VT == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(c1));
        else if (VT == 6) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(c1), "v"(c2));
        else if (VT == 7) {
          const int q = (2 * j) & 31;
          f32x2 t = {v[q], v[q + 1]};
          const f32x2 k1 = {c1, c1};
          asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t) : "v"(k1));
          v[q] = t[0]; v[q + 1] = t[1];
        } else if (VT == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[r]) : "v"(c1));
        else asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[r]) : "v"(c1));
      }
    }
    for (int i = 0; i < 32; ++i) s += v[i];
  }
  const long long c1t = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && (threadIdx.x & 255) == 0) clk[group] = c1t - c0;
}

template <int VT, int MT>
void run(const char* name, int iters_v, int swap = 0, int prio = 0) {
  float* out; long long* clk;
  const int iters_m = 4000;
  (void)hipMalloc(&out, sizeof(float) * 256 * 512);
  (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double us[4]; long long c[4][2];
  for (int active = 1; active <= 3; ++active) {
    hipLaunchKernelGGL((k<VT, MT>), dim3(256), dim3(512), 0, 0, out, 10, 10, active, clk, swap, prio);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<VT, MT>), dim3(256), dim3(512), 0, 0, out, iters_m, iters_v, active, clk, swap, prio);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    us[active] = ms * 1e3;
    (void)hipMemcpy(c[active], clk, 16, hipMemcpyDeviceToHost);
  }
  printf("%-22s swap %d prio %d %s  M alone %7.1f us (%5.2f clk/MFMA-pass-unit)  V alone %7.1f us (%5.2f clk/instr)  both %7.1f us  [ticks M %lld V %lld]  overlap %4.0f %%\n",
         name, swap, prio, MT == 1 ? "16x16x32" : MT == 0 ? "32x32x16" : MT == 2 ? "32x32x16+28nop" : MT == 3 ? "32x32x16+16nop" : MT == 4 ? "32x32x16+24nop" : MT == 5 ? "32x32x16+26nop" : MT == 6 ? "32x32x16+s_nop0" : MT == 7 ? "32x32x16+s_nop1" : "32x32x16+s_nop3", us[1], (double)c[1][0] / (iters_m * 16.0 * 8), us[2], (double)c[2][1] / (iters_v * 64.0), us[3],
         c[3][0], c[3][1], 100.0 * (us[1] + us[2] - us[3]) / (us[1] < us[2] ? us[1] : us[2]));
  (void)hipFree(out); (void)hipFree(clk);
}

int main() {
  // (a) the MFMA waves are the OLDER waves of each SIMD: every VALU class is starved
  run<0, 0>("v_fma_f32", 8000);
  run<1, 0>("v_pk_fma_f32", 8000);
  run<2, 0>("v_exp_f32", 4000);
  run<3, 0>("v_add_u32", 8000);
  run<5, 0>("v_cvt_pk_bf16_f32", 8000);
  run<6, 0>("v_max3_f32", 8000);
  run<7, 0>("v_pk_add_f32", 8000);
  // (b) the VALU waves are the older ones: they run at ~87 % and the MFMA waves at full speed
  run<0, 0>("v_fma_f32", 8000, 1, 0);
  run<2, 0>("v_exp_f32", 4000, 1, 0);
  run<1, 0>("v_pk_fma_f32", 8000, 1, 0);
  // (c) s_setprio on the (younger) VALU waves / on both groups changes nothing
  run<0, 0>("v_fma_f32", 8000, 0, 1);
  run<0, 0>("v_fma_f32", 8000, 0, 2);
  // (d) MFMA waves older, an s_nop after every MFMA: s_nop 0 is free and useless, s_nop 1 costs 11 % of the
  //     matrix stream and lets single-issue VALU classes through; packed fp32 still does not overlap
  run<0, 6>("v_fma_f32", 8000);
  run<0, 7>("v_fma_f32", 8000);
  run<0, 8>("v_fma_f32", 8000);
  run<2, 7>("v_exp_f32", 4000);
  run<2, 8>("v_exp_f32", 4000);
  run<5, 7>("v_cvt_pk_bf16_f32", 8000);
  run<6, 7>("v_max3_f32", 8000);
  run<3, 7>("v_add_u32", 8000);
  run<1, 7>("v_pk_fma_f32", 8000);
  run<7, 7>("v_pk_add_f32", 8000);
  // (e) 4-pass MFMAs behave the same
  run<0, 1>("v_fma_f32", 8000);
  run<2, 1>("v_exp_f32", 4000);
  return 0;
}
