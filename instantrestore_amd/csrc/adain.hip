// adain.hip - AdaIN statistics / application and the invalid-reference zero fill (gfx950).
//
// adain() of face_replace/models/attn_processors.py:7-18 renormalises every reference V to the
// degraded image's V statistics (call site :242-246).  It is an HBM-streaming op: here the
// statistics are ONE pass over V_self and the N reference V's (each element read exactly once,
// 16-byte loads, 128-byte head rows), merged in fp32 with Chan's parallel-variance formula, and
// the result is emitted as a per-(b, n, head, channel) affine  x*a + b  that the attention kernel
// folds into its V staging - the renormalised V is never written to memory.
#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int ROWS = IR_ADAIN_ROWS;  // token rows per workgroup
constexpr int AT = 256;              // threads: 32 row slots x 8 sixteen-byte column slots

// Chan et al. merge of (n, mean, M2) partials.
// a = (sigma_v + eps) / (sigma_x + eps) (attn_processors.py:10,16,245) - except where the content std is EXACTLY 0 (M2 == 0:
// every token of the reference channel holds one value; a zero-filled reference, pix2pix_turbo.py:269-273, or a short token
// axis whose values round to the same 16-bit number).  There x - mean(x) is exactly 0 and the reference's adain() returns the
// style mean whatever the ratio, while the affine form a*x + b with a = sigma_v / 1e-5 ~ 1e5 has to cancel a*x against
// -a*mean(x) in fp32 (round-4 soak): the fused result lost the style mean.  Any small ratio gives the same
// function on such a channel; the ideal is a = 0 (what oracle.adain_affine_np reports), but the 64-row attention kernel's
// ratio frame divides by a, so the kernels emit a = 2^-24 (sigma_v + eps): b = mean(V_self) - mean(x) * a then differs from
// the style mean by < 6e-8 |x| sigma_v, and a*x + b from it by the fp32 rounding of that.
__device__ __forceinline__ float adain_scale(float sd_v_eps, float sd_x_eps, bool content_is_constant) {
  return content_is_constant ? sd_v_eps * 5.9604644775390625e-8f : sd_v_eps / sd_x_eps;
}

__device__ __forceinline__ void chan_merge(float& n, float& mean, float& m2, float nb, float meanb, float m2b) {
  if (nb == 0.f) return;
  const float nt = n + nb;
  const float delta = meanb - mean;
  const float f = nb / nt;
  mean = mean + delta * f;
  m2 = m2 + m2b + delta * delta * n * f;
  n = nt;
}

// grid: (nchunk, H, B*(1+N)); one workgroup reduces ROWS tokens of one head of one matrix.
template <typename T>
__global__ void __launch_bounds__(AT) adain_partial_kernel(const AdainKParams p) {
  using v8 = typename ElemTraits<T>::v8;
  __shared__ float red[32][8][17];  // [row slot][col slot][n, mean[8], m2[8]]

  const int mat = blockIdx.z;
  const int h = blockIdx.y;
  const int chunk = blockIdx.x;
  const int b = mat / (1 + p.N);
  const int j = mat - b * (1 + p.N);  // 0 = self V, 1.. = reference j-1
  const T* base;
  int64_t sl;
  int len;
  if (j == 0) {
    base = (const T*)p.v_self + (int64_t)b * p.vs_sb + (int64_t)h * p.vs_sh;
    sl = p.vs_sl; len = p.Ls;
  } else {
    base = (const T*)p.v_ref + (int64_t)b * p.vr_sb + (int64_t)(j - 1) * p.vr_sn + (int64_t)h * p.vr_sh;
    sl = p.vr_sl; len = p.Lr;
  }
  const int r_begin = chunk * ROWS;
  const int r_end = (r_begin + ROWS < len) ? r_begin + ROWS : len;
  float* wsp = p.ws + (((int64_t)mat * p.H + h) * p.nchunk + chunk) * 128;

  const int tid = threadIdx.x;
  const int slot = tid & 7;
  const int rs = tid >> 3;

  // shifted single-pass sums: K = first value seen by this thread (kills cancellation)
  float cnt = 0.f, K[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { K[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
  constexpr int PER = ROWS / 32;
  v8 xs[PER];
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    int r = r_begin + rs + it * 32;
    const int rc = r < r_end ? r : (r_end - 1 > 0 ? r_end - 1 : 0);
    xs[it] = *(const v8*)(base + (int64_t)rc * sl + slot * 8);
  }
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int r = r_begin + rs + it * 32;
    if (r < r_end) {
      const f32x8 f = __builtin_convertvector(xs[it], f32x8);
      if (cnt == 0.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) K[i] = f[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = f[i] - K[i];
        s1[i] += d;
        s2[i] = __builtin_fmaf(d, d, s2[i]);
      }
      cnt += 1.f;
    }
  }
  {
    float* o = &red[rs][slot][0];
    o[0] = cnt;
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[1 + i] = K[i] + s1[i] * inv;
      o[9 + i] = s2[i] - s1[i] * s1[i] * inv;
    }
  }
  __syncthreads();
  if (tid < 64) {  // one thread per channel merges the 32 row-slot partials
    const int cs = tid >> 3, ci = tid & 7;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int k = 0; k < 32; ++k) {
      const float* o = &red[k][cs][0];
      chan_merge(n, mean, m2, o[0], o[1 + ci], o[9 + ci]);
    }
    wsp[tid] = mean;
    wsp[64 + tid] = m2;
  }
}

// grid: (B*N*H); 64 threads = channels. Merges chunk partials, emits the affine.
__global__ void __launch_bounds__(64) adain_finalize_kernel(const AdainKParams p) {
  const int d = threadIdx.x;
  int idx = blockIdx.x;
  const int h = idx % p.H; idx /= p.H;
  const int n = idx % p.N;
  const int b = idx / p.N;

  auto total = [&](int mat, int len, float& mean, float& m2) {
    float cn = 0.f; mean = 0.f; m2 = 0.f;
    const int nch = (len + ROWS - 1) / ROWS;
    const float* w = p.ws + (((int64_t)mat * p.H + h) * p.nchunk) * 128;
    for (int c = 0; c < nch; ++c) {
      const int rows = (c + 1) * ROWS <= len ? ROWS : len - c * ROWS;
      chan_merge(cn, mean, m2, (float)rows, w[c * 128 + d], w[c * 128 + 64 + d]);
    }
  };
  float mu_v, m2_v, mu_x, m2_x;
  total(b * (1 + p.N), p.Ls, mu_v, m2_v);
  total(b * (1 + p.N) + 1 + n, p.Lr, mu_x, m2_x);
  // torch.std default: unbiased (n-1); a single token gives 0/0 = NaN exactly like torch
  const float sd_v = sqrtf(m2_v / (float)(p.Ls - 1)) + p.eps;
  const float sd_x = sqrtf(m2_x / (float)(p.Lr - 1)) + p.eps;
  const float a = adain_scale(sd_v, sd_x, m2_x == 0.f);
  const int64_t o = (((int64_t)b * p.N + n) * p.H + h) * 64 + d;
  p.a[o] = a;
  p.b[o] = mu_v - mu_x * a;
}

// Same result, one workgroup per (b, h): the (1+N) x nchunk partials are first pulled into LDS with
// independent coalesced loads (the per-channel loop above waits for a fresh cache miss per chunk), then
// merged from LDS.  Used when they fit in `lds_floats` of dynamic shared memory.
__global__ void __launch_bounds__(256) adain_finalize_staged_kernel(const AdainKParams p) {
  extern __shared__ __attribute__((aligned(16))) float fsm[];
  const int tid = threadIdx.x;
  const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
  const int per_mat = p.nchunk * 128;
  for (int j = 0; j <= p.N; ++j) {
    const float* g = p.ws + (((int64_t)(b * (1 + p.N) + j) * p.H + h) * p.nchunk) * 128;
    for (int i = tid * 4; i < per_mat; i += 256 * 4) *(f32x4*)&fsm[j * per_mat + i] = *(const f32x4*)&g[i];
  }
  __syncthreads();
  float* stats = fsm + (1 + p.N) * per_mat;   // [(1+N)][mean 64 | M2 64]
  for (int i = tid; i < (1 + p.N) * 64; i += 256) {
    const int j = i >> 6, d = i & 63;
    const int len = j == 0 ? p.Ls : p.Lr;
    const int nch = (len + ROWS - 1) / ROWS;
    float cn = 0.f, mean = 0.f, m2 = 0.f;
    for (int c = 0; c < nch; ++c) {
      const int rows = (c + 1) * ROWS <= len ? ROWS : len - c * ROWS;
      chan_merge(cn, mean, m2, (float)rows, fsm[j * per_mat + c * 128 + d], fsm[j * per_mat + c * 128 + 64 + d]);
    }
    stats[j * 128 + d] = mean;
    stats[j * 128 + 64 + d] = m2;
  }
  __syncthreads();
  for (int i = tid; i < p.N * 64; i += 256) {
    const int n = i >> 6, d = i & 63;
    const float sd_v = sqrtf(stats[64 + d] / (float)(p.Ls - 1)) + p.eps;
    const float sd_x = sqrtf(stats[(1 + n) * 128 + 64 + d] / (float)(p.Lr - 1)) + p.eps;
    const float a = adain_scale(sd_v, sd_x, stats[(1 + n) * 128 + 64 + d] == 0.f);
    const int64_t o = (((int64_t)b * p.N + n) * p.H + h) * 64 + d;
    p.a[o] = a;
    p.b[o] = stats[d] - stats[(1 + n) * 128 + d] * a;
  }
}

// ---- AdaIN affine from CACHED content statistics (round 3) ---------------------------------------------------------
// The content statistics (mu_x, sigma_x) of a reference V are constant per identity: the K/V-capture layer emits them
// once (ir_token_stats: the same partial kernel and chunk merge as above, so the same bits) and the shared layer reads
// only V_self - 1/(N+1) of the bytes - to finalise a = (sigma_v + eps) / (sigma_x + eps), b = mu_v - mu_x * a.
// Self partials: adain_partial_kernel over the B self matrices alone (ws layout [b][h][chunk][128]).
// grid: (B*H); 256 threads; dynamic LDS: nchunk * 128 floats of partials + 128 of the merged self statistics.
__global__ void __launch_bounds__(256) adain_finalize_cached_kernel(const AdainKParams p) {
  extern __shared__ __attribute__((aligned(16))) float csm[];
  const int tid = threadIdx.x;
  const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
  const int nch = (p.Ls + ROWS - 1) / ROWS;
  const int per_mat = p.nchunk * 128;
  const float* g = p.ws + (((int64_t)b * p.H + h) * p.nchunk) * 128;
  for (int i = tid * 4; i < nch * 128; i += 256 * 4) *(f32x4*)&csm[i] = *(const f32x4*)&g[i];
  __syncthreads();
  float* stats = csm + per_mat;   // mean[64] | M2[64] of V_self
  if (tid < 64) {
    const int d = tid;
    float cn = 0.f, mean = 0.f, m2 = 0.f;
    for (int c = 0; c < nch; ++c) {
      const int rows = (c + 1) * ROWS <= p.Ls ? ROWS : p.Ls - c * ROWS;
      chan_merge(cn, mean, m2, (float)rows, csm[c * 128 + d], csm[c * 128 + 64 + d]);
    }
    stats[d] = mean;
    stats[64 + d] = m2;
  }
  __syncthreads();
  for (int i = tid; i < p.N * 64; i += 256) {
    const int n = i >> 6, d = i & 63;
    const int64_t o = (((int64_t)b * p.N + n) * p.H + h) * 64 + d;
    const float sd_v = sqrtf(stats[64 + d] / (float)(p.Ls - 1)) + p.eps;
    const float sd_x = p.cstd[o] + p.eps;     // cstd = sqrtf(M2_x / (Lr - 1)): the expression of the uncached kernels
    const float a = adain_scale(sd_v, sd_x, p.cstd[o] == 0.f);
    p.a[o] = a;
    p.b[o] = stats[d] - p.cmean[o] * a;
  }
}

// The same without the LDS staging of the partials, for token axes whose chunk count does not fit it (len_self > ~30 000:
// the 256x256-token layers of a 2048 px input): grid (B*H), 64 threads = channels, partials merged straight from memory.
__global__ void __launch_bounds__(64) adain_finalize_cached_direct_kernel(const AdainKParams p) {
  const int d = threadIdx.x;
  const int h = blockIdx.x % p.H, b = blockIdx.x / p.H;
  const int nch = (p.Ls + ROWS - 1) / ROWS;
  const float* g = p.ws + (((int64_t)b * p.H + h) * p.nchunk) * 128;
  float cn = 0.f, mean = 0.f, m2 = 0.f;
  for (int c = 0; c < nch; ++c) {
    const int rows = (c + 1) * ROWS <= p.Ls ? ROWS : p.Ls - c * ROWS;
    chan_merge(cn, mean, m2, (float)rows, g[c * 128 + d], g[c * 128 + 64 + d]);
  }
  const float sd_v = sqrtf(m2 / (float)(p.Ls - 1)) + p.eps;
  for (int n = 0; n < p.N; ++n) {
    const int64_t o = (((int64_t)b * p.N + n) * p.H + h) * 64 + d;
    const float a = adain_scale(sd_v, p.cstd[o] + p.eps, p.cstd[o] == 0.f);
    p.a[o] = a;
    p.b[o] = mean - p.cmean[o] * a;
  }
}

// One-chunk token axes (the 16x16-token class): one workgroup per (b, h) reads V_self and emits the affine. grid: (H, B).
template <typename T>
__global__ void __launch_bounds__(AT) adain_self_small_kernel(const AdainKParams p) {
  using v8 = typename ElemTraits<T>::v8;
  __shared__ float red[32][8][17];
  __shared__ float stats[128];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, slot = tid & 7, rs = tid >> 3;
  constexpr int PER = ROWS / 32;
  const T* base = (const T*)p.v_self + (int64_t)b * p.vs_sb + (int64_t)h * p.vs_sh;
  v8 xs[PER];
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int r = rs + it * 32;
    xs[it] = *(const v8*)(base + (int64_t)(r < p.Ls ? r : p.Ls - 1) * p.vs_sl + slot * 8);
  }
  float cnt = 0.f, K[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { K[i] = 0.f; s1[i] = 0.f; s2[i] = 0.f; }
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    if (rs + it * 32 < p.Ls) {
      const f32x8 f = __builtin_convertvector(xs[it], f32x8);
      if (cnt == 0.f) {
#pragma unroll
        for (int i = 0; i < 8; ++i) K[i] = f[i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = f[i] - K[i];
        s1[i] += d;
        s2[i] = __builtin_fmaf(d, d, s2[i]);
      }
      cnt += 1.f;
    }
  }
  {
    float* o = &red[rs][slot][0];
    o[0] = cnt;
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      o[1 + i] = K[i] + s1[i] * inv;
      o[9 + i] = s2[i] - s1[i] * s1[i] * inv;
    }
  }
  __syncthreads();
  if (tid < 64) {
    const int cs = tid >> 3, ci = tid & 7;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int k = 0; k < 32; ++k) {
      const float* o = &red[k][cs][0];
      chan_merge(n, mean, m2, o[0], o[1 + ci], o[9 + ci]);
    }
    stats[tid] = mean;
    stats[64 + tid] = m2;
  }
  __syncthreads();
  for (int i = tid; i < p.N * 64; i += AT) {
    const int n = i >> 6, d = i & 63;
    const int64_t o = (((int64_t)b * p.N + n) * p.H + h) * 64 + d;
    const float sd_v = sqrtf(stats[64 + d] / (float)(p.Ls - 1)) + p.eps;
    const float sd_x = p.cstd[o] + p.eps;
    const float a = adain_scale(sd_v, sd_x, p.cstd[o] == 0.f);
    p.a[o] = a;
    p.b[o] = stats[d] - p.cmean[o] * a;
  }
}

// grid: (B*(1+N)*H); 64 threads. Plain token statistics (mean, unbiased std) of every matrix.
__global__ void __launch_bounds__(64) token_stats_finalize_kernel(const AdainKParams p) {
  const int d = threadIdx.x;
  int idx = blockIdx.x;
  const int h = idx % p.H;
  const int mat = idx / p.H;
  const int j = mat % (1 + p.N);
  const int len = (j == 0) ? p.Ls : p.Lr;
  float cn = 0.f, mean = 0.f, m2 = 0.f;
  const int nch = (len + ROWS - 1) / ROWS;
  const float* w = p.ws + (((int64_t)mat * p.H + h) * p.nchunk) * 128;
  for (int c = 0; c < nch; ++c) {
    const int rows = (c + 1) * ROWS <= len ? ROWS : len - c * ROWS;
    chan_merge(cn, mean, m2, (float)rows, w[c * 128 + d], w[c * 128 + 64 + d]);
  }
  const int64_t o = ((int64_t)mat * p.H + h) * 64 + d;
  p.a[o] = mean;
  p.b[o] = sqrtf(m2 / (float)(len - 1));
}

// y = x*a + b over (B,N,L,H,64); one thread = one 16-byte slot, grid-stride over head rows.
template <typename T>
__global__ void __launch_bounds__(256) adain_apply_kernel(const AdainApplyKParams p) {
  using v8 = typename ElemTraits<T>::v8;
  const int slot = threadIdx.x & 7;
  const int64_t rows = (int64_t)p.B * p.N * p.L * p.H;
  for (int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); row < rows; row += (int64_t)gridDim.x * 32) {
    int64_t t = row;
    const int h = (int)(t % p.H); t /= p.H;
    const int l = (int)(t % p.L); t /= p.L;
    const int n = (int)(t % p.N);
    const int b = (int)(t / p.N);
    const T* x = (const T*)p.x + (int64_t)b * p.x_sb + (int64_t)n * p.x_sn + (int64_t)l * p.x_sl + (int64_t)h * p.x_sh + slot * 8;
    T* y = (T*)p.y + (int64_t)b * p.y_sb + (int64_t)n * p.y_sn + (int64_t)l * p.y_sl + (int64_t)h * p.y_sh + slot * 8;
    const int64_t ao = (((int64_t)b * p.N + n) * p.H + h) * 64 + slot * 8;
    f32x8 f = __builtin_convertvector(*(const v8*)x, f32x8);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = __builtin_fmaf(f[i], p.a[ao + i], p.b[ao + i]);
    *(v8*)y = __builtin_convertvector(f, v8);
  }
}

// zero K and V of references n >= valid[b] (pix2pix_turbo.py:269-273); grid (row chunks, N, B)
__global__ void __launch_bounds__(256) zero_refs_kernel(const ZeroRefsKParams p) {
  const int b = blockIdx.z, n = blockIdx.y;
  if (n < p.valid[b]) return;
  const int slot = threadIdx.x & 7;
  const int64_t rows = (int64_t)p.L * p.H;
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); row < rows; row += (int64_t)gridDim.x * 32) {
    const int h = (int)(row % p.H);
    const int64_t l = row / p.H;
    unsigned short* k = (unsigned short*)p.k + (int64_t)b * p.k_sb + (int64_t)n * p.k_sn + l * p.k_sl + (int64_t)h * p.k_sh + slot * 8;
    unsigned short* v = (unsigned short*)p.v + (int64_t)b * p.v_sb + (int64_t)n * p.v_sn + l * p.v_sl + (int64_t)h * p.v_sh + slot * 8;
    *(uint4*)k = z;
    *(uint4*)v = z;
  }
}

}  // namespace

hipError_t ir_launch_adain_stats(const AdainKParams& p, int dtype, hipStream_t s) {
  // (round 3: a one-launch kernel for the 16x16-token layers - every matrix of a (b, h) pair fetched up front by ONE workgroup -
  //  was 22.9 us against 11.1 us for the two launches below once the step is replayed from a hipGraph: five reductions and
  //  their 32-step merges in sequence, on 160 workgroups; tools/gpu_adain_time.py)
  const dim3 grid(p.nchunk, p.H, p.B * (1 + p.N));
  if (dtype == 1) hipLaunchKernelGGL((adain_partial_kernel<__bf16>), grid, dim3(AT), 0, s, p);
  else hipLaunchKernelGGL((adain_partial_kernel<_Float16>), grid, dim3(AT), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const size_t lds = (size_t)(1 + p.N) * (p.nchunk * 128 + 128) * sizeof(float);
  // (short token axes have one or two chunks: nothing to overlap, and more, smaller workgroups win)
  if (p.nchunk >= 8 && lds <= 60 * 1024) hipLaunchKernelGGL(adain_finalize_staged_kernel, dim3(p.B * p.H), dim3(256), lds, s, p);
  else hipLaunchKernelGGL(adain_finalize_kernel, dim3(p.B * p.N * p.H), dim3(64), 0, s, p);
  return hipGetLastError();
}

hipError_t ir_launch_token_stats(const AdainKParams& p, int dtype, hipStream_t s) {
  const dim3 grid(p.nchunk, p.H, p.B * (1 + p.N));
  if (dtype == 1) hipLaunchKernelGGL((adain_partial_kernel<__bf16>), grid, dim3(AT), 0, s, p);
  else hipLaunchKernelGGL((adain_partial_kernel<_Float16>), grid, dim3(AT), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(token_stats_finalize_kernel, dim3(p.B * (1 + p.N) * p.H), dim3(64), 0, s, p);
  return hipGetLastError();
}

hipError_t ir_launch_adain_stats_cached(const AdainKParams& p, int dtype, hipStream_t s) {
  if (p.nchunk == 1) {
    if (dtype == 1) hipLaunchKernelGGL((adain_self_small_kernel<__bf16>), dim3(p.H, p.B), dim3(AT), 0, s, p);
    else hipLaunchKernelGGL((adain_self_small_kernel<_Float16>), dim3(p.H, p.B), dim3(AT), 0, s, p);
    return hipGetLastError();
  }
  AdainKParams q = p;
  q.N = 0;                                   // the partial pass sees B matrices: the self V's
  const dim3 grid(p.nchunk, p.H, p.B);
  if (dtype == 1) hipLaunchKernelGGL((adain_partial_kernel<__bf16>), grid, dim3(AT), 0, s, q);
  else hipLaunchKernelGGL((adain_partial_kernel<_Float16>), grid, dim3(AT), 0, s, q);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const size_t lds = (size_t)(p.nchunk * 128 + 128) * sizeof(float);
  if (lds <= 60 * 1024) hipLaunchKernelGGL(adain_finalize_cached_kernel, dim3(p.B * p.H), dim3(256), lds, s, p);
  else hipLaunchKernelGGL(adain_finalize_cached_direct_kernel, dim3(p.B * p.H), dim3(64), 0, s, p);   // same merge order, same bits
  return hipGetLastError();
}

hipError_t ir_launch_adain_apply(const AdainApplyKParams& p, int dtype, hipStream_t s) {
  const int64_t rows = (int64_t)p.B * p.N * p.L * p.H;
  int64_t blocks = (rows + 31) / 32;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (dtype == 1) hipLaunchKernelGGL((adain_apply_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((adain_apply_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, s, p);
  return hipGetLastError();
}

hipError_t ir_launch_zero_refs(const ZeroRefsKParams& p, hipStream_t s) {
  const int64_t rows = (int64_t)p.L * p.H;
  int64_t bx = (rows + 31) / 32;
  if (bx > 256) bx = 256;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(zero_refs_kernel, dim3((unsigned)bx, p.N, p.B), dim3(256), 0, s, p);
  return hipGetLastError();
}

// ---- tensor2im: (B,3,H,W) in [-1,1] -> uint8 (B,H,W,3), the output path of the caller -------------
// face_replace/training/utils/vis_utils.py:14-23 (tensor2im(var, unnorm=True), called at
// inference/test.py:139): var*0.5 + 0.5 IN THE TENSOR'S DTYPE, clamp to [0,1], *255 in that dtype,
// truncation to uint8, CHW -> HWC.  Done on the device so only H*W*3 bytes cross PCIe.  Every
// intermediate is rounded to T exactly where the reference rounds, so the bytes are bit-identical.
namespace {
template <typename T>
__global__ void __launch_bounds__(256) tensor2im_kernel(const T* __restrict__ x, unsigned char* __restrict__ out,
                                                        int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                                        int B, int C, int H, int W) {
  const int64_t n = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int w = (int)(i % W);
    const int64_t t = i / W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const T* px = x + b * sb + h * sh + w * sw;
    unsigned char* po = out + i * C;
    for (int c = 0; c < C; ++c) {
      T v = px[c * sc];
      v = (T)((float)v * 0.5f);          // var *= 0.5   (exact in binary floating point)
      v = (T)((float)v + 0.5f);          // var += 0.5   (one rounding to T)
      float f = (float)v;
      f = f < 0.f ? 0.f : (f > 1.f ? 1.f : f);
      v = (T)(f * 255.0f);               // var *= 255   (one rounding to T)
      po[c] = (unsigned char)(int)(float)v;  // astype('uint8'): truncation
    }
  }
}
}  // namespace

hipError_t ir_launch_tensor2im(const void* x, void* out, int dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                               int B, int C, int H, int W, hipStream_t s) {
  const int64_t n = (int64_t)B * H * W;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  const dim3 g((unsigned)blocks), t(256);
  if (dtype == 0) hipLaunchKernelGGL((tensor2im_kernel<_Float16>), g, t, 0, s, (const _Float16*)x, (unsigned char*)out, sb, sc, sh, sw, B, C, H, W);
  else if (dtype == 1) hipLaunchKernelGGL((tensor2im_kernel<__bf16>), g, t, 0, s, (const __bf16*)x, (unsigned char*)out, sb, sc, sh, sw, B, C, H, W);
  else hipLaunchKernelGGL((tensor2im_kernel<float>), g, t, 0, s, (const float*)x, (unsigned char*)out, sb, sc, sh, sw, B, C, H, W);
  return hipGetLastError();
}

// ---- round 4: AdaIN from the partial statistics the projection GEMMs leave behind (ir_colstats.h) -----------------------
// The q/k/v projection of a shared layer stores the partials of V_self (style), the projection of the K/V-capture layer
// those of every reference V (content), one (mean[64], M2[64]) per (row block, head), row blocks of `rows` tokens in
// token order, so a matrix's partials are consecutive: ws[(set * len / rows + c) * H + h][128].  No pass over V is left.
namespace {

// (mean, M2) of a matrix from its `nch` partials of `rows` tokens each, by the 1024 threads of a workgroup: thread (g, d),
// g = tid >> 6 in [0, 16), holds chunks g, g + 16, ... of channel d in registers (all loads in flight at once: the kernel
// is one memory latency long).  Equal counts make the merge a two-pass sum with no serial chain:
//   mean = (1 / nch) * sum_i mean_i,     M2 = sum_i [ M2_i + rows * (mean_i - mean)^2 ]
// (first version: Chan partials merged one after another by one thread per channel - 64 dependent divisions at the
// 64 x 64-token class, 9.7 us per launch; second: 256 threads re-reading memory in both passes, 16 us).  The sums over g
// run in a fixed order: deterministic.
constexpr int kMergePerMax = 16;   // chunks per thread held in registers: nch <= 256 (16 384 tokens in 64-row blocks)

template <int PER>
struct MergeRegs {
  float m[PER], q[PER];
};

// every load unconditional (out-of-range chunks re-read the last one and are weighted out): all 2 PER loads in flight at once
template <int PER>
__device__ __forceinline__ void merge_load(MergeRegs<PER>& r, const float* __restrict__ w, int64_t stride, int nch, int g, int d) {
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = g + 16 * i;
    const int cc = c < nch ? c : (nch > 0 ? nch - 1 : 0);
    r.m[i] = w[cc * stride + d];
    r.q[i] = w[cc * stride + 64 + d];
  }
}
template <int PER>
__device__ __forceinline__ float merge_sum_mean(const MergeRegs<PER>& r, int nch, int g) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) s += (g + 16 * i < nch) ? r.m[i] : 0.f;
  return s;
}
template <int PER>
__device__ __forceinline__ float merge_sum_m2(const MergeRegs<PER>& r, int nch, int g, float rows, float mean) {
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const float dm = r.m[i] - mean;
    q += (g + 16 * i < nch) ? (r.q[i] + rows * dm * dm) : 0.f;
  }
  return q;
}
__device__ __forceinline__ float sum16(const float* red, int d) {   // over the 16 groups, fixed order
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < 16; ++g) s += red[g * 64 + d];
  return s;
}

// grid: (B*N*H); 1024 threads.  One workgroup per (b, n, h): style (V_self of b) + content (reference n) -> a, b.
template <int PER>
__global__ void __launch_bounds__(1024) adain_affine_partials_kernel(const AdainPartialsKParams p) {
  __shared__ float red[4][1024];
  const int tid = threadIdx.x, d = tid & 63, g = tid >> 6;
  int idx = blockIdx.x;
  const int h = idx % p.H; idx /= p.H;
  const int n = idx % p.N, b = idx / p.N;
  const int64_t hs = (int64_t)p.H * 128;
  const int nch_s = p.Ls / p.style_rows;
  const bool zeroed = p.valid != nullptr && n >= p.valid[b];       // zero-filled reference (pix2pix_turbo.py:269-273): statistics (0, 0)
  const bool merge_c = p.content_ws != nullptr && !zeroed;
  const int nch_c = merge_c ? p.Lr / p.content_rows : 0;
  MergeRegs<PER> rs, rc;
  merge_load<PER>(rs, p.style_ws + ((int64_t)b * nch_s) * hs + h * 128, hs, nch_s, g, d);
  merge_load<PER>(rc, merge_c ? p.content_ws + ((int64_t)(b * p.N + n) * nch_c) * hs + h * 128 : p.style_ws + ((int64_t)b * nch_s) * hs + h * 128,
                  hs, merge_c ? nch_c : 1, g, d);
  red[0][tid] = merge_sum_mean<PER>(rs, nch_s, g);
  red[1][tid] = merge_sum_mean<PER>(rc, nch_c, g);
  __syncthreads();
  const float mean_s = sum16(red[0], d) / (float)nch_s;
  const float mean_c = merge_c ? sum16(red[1], d) / (float)nch_c : 0.f;
  red[2][tid] = merge_sum_m2<PER>(rs, nch_s, g, (float)p.style_rows, mean_s);
  red[3][tid] = merge_sum_m2<PER>(rc, nch_c, g, (float)p.content_rows, mean_c);
  __syncthreads();
  if (g == 0) {
    const int64_t o = (((int64_t)b * p.N + n) * p.H + h) * 64 + d;
    float mu_x, sd_x;
    if (zeroed) { mu_x = 0.f; sd_x = 0.f; }
    else if (merge_c) { mu_x = mean_c; sd_x = sqrtf(sum16(red[3], d) / (float)(p.Lr - 1)); }
    else { mu_x = p.cmean[o]; sd_x = p.cstd[o]; }
    const float sd_v = sqrtf(sum16(red[2], d) / (float)(p.Ls - 1)) + p.eps;
    const float a = adain_scale(sd_v, sd_x + p.eps, sd_x == 0.f);
    p.a[o] = a;
    p.b[o] = mean_s - mu_x * a;
  }
}

// grid: (nsets*H); 1024 threads.  mean / unbiased std of every matrix from its partials.
template <int PER>
__global__ void __launch_bounds__(1024) token_stats_partials_kernel(const float* ws, int rows, int H, int len, float* mean_out, float* std_out) {
  __shared__ float red[2][1024];
  const int tid = threadIdx.x, d = tid & 63, g = tid >> 6;
  const int h = blockIdx.x % H, set = blockIdx.x / H;
  const int nch = len / rows;
  MergeRegs<PER> r;
  merge_load<PER>(r, ws + ((int64_t)set * nch) * H * 128 + h * 128, (int64_t)H * 128, nch, g, d);
  red[0][tid] = merge_sum_mean<PER>(r, nch, g);
  __syncthreads();
  const float mean = sum16(red[0], d) / (float)nch;
  red[1][tid] = merge_sum_m2<PER>(r, nch, g, (float)rows, mean);
  __syncthreads();
  if (g == 0) {
    const int64_t o = ((int64_t)set * H + h) * 64 + d;
    mean_out[o] = mean;
    std_out[o] = sqrtf(sum16(red[1], d) / (float)(len - 1));
  }
}

}  // namespace

hipError_t ir_launch_adain_affine_partials(const AdainPartialsKParams& p, hipStream_t s) {
  const int ns = p.Ls / p.style_rows, nc = p.content_ws != nullptr ? p.Lr / p.content_rows : 1;
  const int nch = ns > nc ? ns : nc;                       // chunks per thread: 1 (<= 16 partials), 4 (<= 64), 16 (<= 256)
  const dim3 grid(p.B * p.N * p.H), blk(1024);
  if (nch <= 16) hipLaunchKernelGGL(adain_affine_partials_kernel<1>, grid, blk, 0, s, p);
  else if (nch <= 64) hipLaunchKernelGGL(adain_affine_partials_kernel<4>, grid, blk, 0, s, p);
  else hipLaunchKernelGGL(adain_affine_partials_kernel<kMergePerMax>, grid, blk, 0, s, p);
  return hipGetLastError();
}

hipError_t ir_launch_token_stats_partials(const float* ws, int rows, int nsets, int H, int len, float* mean, float* std, hipStream_t s) {
  const int nch = len / rows;
  const dim3 grid(nsets * H), blk(1024);
  if (nch <= 16) hipLaunchKernelGGL(token_stats_partials_kernel<1>, grid, blk, 0, s, ws, rows, H, len, mean, std);
  else if (nch <= 64) hipLaunchKernelGGL(token_stats_partials_kernel<4>, grid, blk, 0, s, ws, rows, H, len, mean, std);
  else hipLaunchKernelGGL(token_stats_partials_kernel<kMergePerMax>, grid, blk, 0, s, ws, rows, H, len, mean, std);
  return hipGetLastError();
}

int ir_adain_partials_max_chunks(void) { return 16 * kMergePerMax; }
