/* The drop-in boundary used from plain C: no PyTorch, no Python - only libinstantrestore_hip.so,
 * include/instantrestore_hip.h and the HIP runtime for device memory.  Runs one shared-attention call
 * (self + 2 references, AdaIN on, bf16) and checks it against a float64 loop written here.
 *
 *   gcc -std=gnu11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Linstantrestore_amd -linstantrestore_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/instantrestore_amd -Wl,-rpath,/opt/rocm/lib -lm -o /tmp/c_abi_demo && /tmp/c_abi_demo
 *   (-D__HIP_PLATFORM_AMD__ is what the vendor's hip_runtime_api.h wants from a plain C compiler)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "instantrestore_hip.h"

#define B 1
#define H 2
#define L 96
#define N 2
#define C (H * 64)

static uint16_t f2bf(float f) { /* round to nearest even */
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static float frand(void) { return (float)rand() / RAND_MAX * 2.f - 1.f; }

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define CHECK_IR(x) do { int r_ = (x); if (r_ != IR_OK) { printf("IR error %d: %s (line %d)\n", r_, ir_last_error_string(), __LINE__); return 3; } } while (0)

int main(void) {
  if (ir_abi_version() != IR_ABI_VERSION) { printf("ABI mismatch\n"); return 1; }
  printf("%s\n", ir_build_info());
  srand(7);
  const size_t nt = (size_t)B * L * C, nr = (size_t)B * N * L * C;
  uint16_t *q = malloc(2 * nt), *k = malloc(2 * nt), *v = malloc(2 * nt), *rk = malloc(2 * nr), *rv = malloc(2 * nr), *out = malloc(2 * nt);
  for (size_t i = 0; i < nt; ++i) { q[i] = f2bf(frand()); k[i] = f2bf(frand()); v[i] = f2bf(frand() * 0.8f + 0.2f); }
  for (size_t i = 0; i < nr; ++i) { rk[i] = f2bf(frand()); rv[i] = f2bf(frand() * 1.3f - 0.4f); }

  void *dq, *dk, *dv, *drk, *drv, *dout, *dws, *dws2; float *da, *db;
  CHECK_HIP(hipMalloc(&dq, 2 * nt)); CHECK_HIP(hipMalloc(&dk, 2 * nt)); CHECK_HIP(hipMalloc(&dv, 2 * nt));
  CHECK_HIP(hipMalloc(&drk, 2 * nr)); CHECK_HIP(hipMalloc(&drv, 2 * nr)); CHECK_HIP(hipMalloc(&dout, 2 * nt));
  CHECK_HIP(hipMalloc((void**)&da, sizeof(float) * B * N * C)); CHECK_HIP(hipMalloc((void**)&db, sizeof(float) * B * N * C));
  const size_t ws_stats = ir_adain_stats_workspace_bytes(B, H, L, N, L), ws_attn = ir_shared_attn_workspace_bytes();
  CHECK_HIP(hipMalloc(&dws, ws_stats)); CHECK_HIP(hipMalloc(&dws2, ws_attn));
  CHECK_HIP(hipMemcpy(dq, q, 2 * nt, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dk, k, 2 * nt, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dv, v, 2 * nt, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(drk, rk, 2 * nr, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(drv, rv, 2 * nr, hipMemcpyHostToDevice));

  /* AdaIN affine from token statistics (attn_processors.py:9-10, 244-245) */
  CHECK_IR(ir_adain_stats(IR_DTYPE_BF16, B, H, L, N, L, dv, (int64_t)L * C, C, 64, drv, (int64_t)N * L * C, (int64_t)L * C, C, 64,
                          1e-5f, da, db, dws, ws_stats, NULL));
  /* fused extended self-attention (attn_processors.py:232-264) */
  ir_shared_attn_args a; memset(&a, 0, sizeof(a));
  a.struct_size = sizeof(a); a.dtype = IR_DTYPE_BF16; a.flags = IR_FLAG_INCLUDE_SELF;
  a.batch = B; a.heads = H; a.len_q = L; a.len_self = L; a.n_refs = N; a.len_ref = L; a.scale = 0.125f;
  a.q = dq; a.k_self = dk; a.v_self = dv; a.k_ref = drk; a.v_ref = drv; a.adain_a = da; a.adain_b = db; a.out = dout;
  a.q_sb = a.ks_sb = a.vs_sb = a.o_sb = (int64_t)L * C; a.q_sl = a.ks_sl = a.vs_sl = a.o_sl = C; a.q_sh = a.ks_sh = a.vs_sh = a.o_sh = 64;
  a.kr_sb = a.vr_sb = (int64_t)N * L * C; a.kr_sn = a.vr_sn = (int64_t)L * C; a.kr_sl = a.vr_sl = C; a.kr_sh = a.vr_sh = 64;
  a.workspace = dws2; a.workspace_bytes = ws_attn;
  CHECK_IR(ir_shared_attn_fwd(&a, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(out, dout, 2 * nt, hipMemcpyDeviceToHost));

  /* float64 restatement: per head, AdaIN of every reference V to the self V statistics, softmax over [self, refs] */
  double maxerr = 0, maxref = 0;
  for (int h = 0; h < H; ++h) {
    double mu_s[64], sd_s[64], aa[N][64], bb[N][64];
    for (int d = 0; d < 64; ++d) {
      double m = 0, s2 = 0;
      for (int l = 0; l < L; ++l) m += bf2f(v[(size_t)l * C + h * 64 + d]);
      m /= L;
      for (int l = 0; l < L; ++l) { double x = bf2f(v[(size_t)l * C + h * 64 + d]) - m; s2 += x * x; }
      mu_s[d] = m; sd_s[d] = sqrt(s2 / (L - 1)) + 1e-5;
      for (int n = 0; n < N; ++n) {
        double mr = 0, sr = 0;
        for (int l = 0; l < L; ++l) mr += bf2f(rv[((size_t)n * L + l) * C + h * 64 + d]);
        mr /= L;
        for (int l = 0; l < L; ++l) { double x = bf2f(rv[((size_t)n * L + l) * C + h * 64 + d]) - mr; sr += x * x; }
        aa[n][d] = sd_s[d] / (sqrt(sr / (L - 1)) + 1e-5);
        bb[n][d] = mu_s[d] - mr * aa[n][d];
      }
    }
    for (int i = 0; i < L; ++i) {
      double s[(1 + N) * L], mx = -1e300, den = 0, o[64] = {0};
      for (int j = 0; j < (1 + N) * L; ++j) {
        const uint16_t* kp = j < L ? &k[(size_t)j * C + h * 64] : &rk[((size_t)(j - L)) * C + h * 64];
        double acc = 0;
        for (int d = 0; d < 64; ++d) acc += (double)bf2f(q[(size_t)i * C + h * 64 + d]) * bf2f(kp[d]);
        s[j] = acc * 0.125; if (s[j] > mx) mx = s[j];
      }
      for (int j = 0; j < (1 + N) * L; ++j) {
        const double pj = exp(s[j] - mx); den += pj;
        for (int d = 0; d < 64; ++d) {
          double vv;
          if (j < L) vv = bf2f(v[(size_t)j * C + h * 64 + d]);
          else { const int n = (j - L) / L; vv = bf2f(rv[((size_t)(j - L)) * C + h * 64 + d]) * aa[n][d] + bb[n][d]; }
          o[d] += pj * vv;
        }
      }
      for (int d = 0; d < 64; ++d) {
        const double ref = o[d] / den, got = bf2f(out[(size_t)i * C + h * 64 + d]);
        if (fabs(ref) > maxref) maxref = fabs(ref);
        if (fabs(ref - got) > maxerr) maxerr = fabs(ref - got);
      }
    }
  }
  const double bound = 8e-3 * (maxref > 1 ? maxref : 1);
  printf("max|err| %.3e (bound %.3e, max|ref| %.3f): %s\n", maxerr, bound, maxref, maxerr <= bound ? "OK" : "FAIL");
  if (maxerr > bound) return 4;

  /* The per-reference attention mass gradio_demo.py:119-127 reduces attention_probs to, without the (B,H,L,Lkv) tensor:
     ABI v9 - a BY-PRODUCT of the attention call itself (args.seg_mass -> fp32 (B, H, L, 1 + N); every row sums to 1);
     ABI v8 - a second pass over Q and K from the call's LSE (ir_attn_segment_mass), for callers that only kept the LSE */
  float *dlse, *dmass, *dmass2;
  const size_t nm = (size_t)B * H * L * (1 + N);
  CHECK_HIP(hipMalloc((void**)&dlse, sizeof(float) * B * H * L)); CHECK_HIP(hipMalloc((void**)&dmass, sizeof(float) * nm));
  CHECK_HIP(hipMalloc((void**)&dmass2, sizeof(float) * nm));
  a.lse = dlse;
  a.seg_mass = dmass;
  CHECK_IR(ir_shared_attn_fwd(&a, NULL));
  a.seg_mass = NULL;
  CHECK_IR(ir_attn_segment_mass(&a, dmass2, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  float* mass = malloc(sizeof(float) * nm);
  float* mass2 = malloc(sizeof(float) * nm);
  CHECK_HIP(hipMemcpy(mass, dmass, sizeof(float) * nm, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(mass2, dmass2, sizeof(float) * nm, hipMemcpyDeviceToHost));
  double worst = 0, apart = 0;
  for (size_t r = 0; r < (size_t)B * H * L; ++r) {
    double sum = 0;
    for (int s2 = 0; s2 <= N; ++s2) {
      sum += mass[r * (1 + N) + s2];
      if (fabs(mass[r * (1 + N) + s2] - mass2[r * (1 + N) + s2]) > apart) apart = fabs(mass[r * (1 + N) + s2] - mass2[r * (1 + N) + s2]);
    }
    if (fabs(sum - 1.0) > worst) worst = fabs(sum - 1.0);
  }
  const int ok = worst <= 1e-5 && apart <= 1e-4;
  printf("segment mass (by-product of the call): max |row sum - 1| %.3e, max |by-product - second pass| %.3e: %s\n", worst, apart, ok ? "OK" : "FAIL");
  return ok ? 0 : 5;
}
