// Microbenchmark (gfx950): how many single-issue VALU instructions fit, for free, into the gap between two
// v_mfma_f32_32x32x16_bf16 of ONE wave, with one or two waves per SIMD?  (The round-1 reading "inside one wave
// nothing overlaps" was taken on same-accumulator chains; MI355X_MICROARCH.md says <= 5 fillers per gap are hidden
// with one wave per SIMD.)  Stream per iteration: 32 MFMAs rotating over 4 accumulators (the attention kernel's
// order), F fillers after each.  KIND 0: v_fma_f32, 1: v_exp_f32, 2: the softmax mix (exp, fma, add, max3, cvt_pk).
// Reports shader cycles per MFMA from s_memtime of wave 0 and wall-clock TFLOP/s (random operands).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int KIND, int J>
__device__ __forceinline__ void filler(float (&v)[32], float c1, float c2) {
  float& x = v[J & 31];
  if constexpr (KIND == 0) {
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
  } else if constexpr (KIND == 1) {
    asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  } else {   // the softmax mix: exp : fma : add : max3 : cvt_pk = 2 : 2 : 2 : 1 : 1
    constexpr int r = J % 8;
    if constexpr (r == 0 || r == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if constexpr (r == 1 || r == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    else if constexpr (r == 2 || r == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
    else if constexpr (r == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    else { unsigned q; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(q) : "v"(x), "v"(c1)); asm volatile("" ::"v"(q)); }
  }
}

template <int KIND, int J, int N>
__device__ __forceinline__ void fillers(float (&v)[32], float c1, float c2) {
  if constexpr (N > 0) {
    filler<KIND, J>(v, c1, c2);
    fillers<KIND, J + 1, N - 1>(v, c1, c2);
  }
}

template <int F, int KIND, int M>
__device__ __forceinline__ void steps(f32x16 (&acc)[4], const bf16x8 (&a)[8], const bf16x8 (&b)[8], float (&v)[32], float c1, float c2) {
  if constexpr (M < 32) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[M & 3]) : "v"(a[M & 7]), "v"(b[(M * 3) & 7]));
    fillers<KIND, M * F, F>(v, c1, c2);
    steps<F, KIND, M + 1>(acc, a, b, v, c1, c2);
  }
}

template <int F, int KIND, int NT>
__global__ void __launch_bounds__(NT, 1) k(float* out, long long* cyc, int iters, int zero) {
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
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    steps<F, KIND, 0>(acc, a, b, v, c1, c2);
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += acc[j][i];
  for (int i = 0; i < 32; ++i) sum += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int F, int KIND, int NT>
void run(int zero) {
  float* out; long long* cyc;
  const int iters = 4000;
  (void)hipMalloc(&out, sizeof(float) * 256 * NT);
  (void)hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<F, KIND, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters, zero);
  (void)hipEventRecord(e0);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<F, KIND, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters, zero);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 32;
  printf("waves/SIMD %d  kind %d  fillers/gap %d  %s: %6.1f ticks/MFMA (wave 0)  %7.1f ns/iter  %6.0f TFLOP/s\n", NT / 256, KIND, F,
         zero ? "zeros " : "random", (double)c / nm, ms * 1e6 / (3.0 * iters), 3.0 * nm * 32768.0 * (NT / 64) * 256 / (ms * 1e-3) / 1e12);
  (void)hipFree(out); (void)hipFree(cyc);
}

template <int KIND, int NT>
void sweep(int zero) {
  run<0, KIND, NT>(zero); run<2, KIND, NT>(zero); run<3, KIND, NT>(zero); run<4, KIND, NT>(zero); run<5, KIND, NT>(zero);
  run<6, KIND, NT>(zero); run<7, KIND, NT>(zero); run<8, KIND, NT>(zero); run<10, KIND, NT>(zero);
}

int main() {
  for (int zero = 1; zero >= 0; --zero) {
    sweep<0, 256>(zero); sweep<2, 256>(zero); sweep<0, 512>(zero); sweep<2, 512>(zero);
    if (zero) { sweep<1, 256>(zero); sweep<1, 512>(zero); }
  }
  return 0;
}
