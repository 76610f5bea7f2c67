// Data formats either side of the hot path, HBM/latency-bound
// byte and elementwise work:
//
//  * the caller's input transform (face_replace/inference/test.py:54-59):
//      Resize(512, LANCZOS) -> CenterCrop(512) -> ToTensor -> Normalize(0.5, 0.5)
//    = Pillow's 8-bit two-pass resampler (pillow==10.4.0, src/libImaging/Resample.c: 22-bit
//    fixed-point taps, the horizontal pass rounded to uint8 before the vertical one) followed by
//    (v/255 - 0.5)/0.5 in float32.  Integer work: the bytes are bit-identical to Pillow's.  Only the
//    crop is computed (the columns of the horizontal pass and the source rows its vertical taps
//    touch), a whole batch of differently sized images per launch pair.
//  * FreeU's skip-feature filter (face_replace/models/unet_2d_condition/block.py:3495-3520 ->
//    diffusers fourier_filter): fft2 -> shift -> scale the (2t)^2 centre bins -> unshift -> ifft2
//    -> real.  Only (2t)^2 DFT bins change, so y = x + (s-1)/(HW) Re sum X(u,v) e^{+i theta}: one
//    wave per (b,c) plane, one read, one write, no FFT, no fp32 round trips through HBM.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "ir_kernels.h"

// ------------------------------------------------------------------------------------------
// host: Pillow's precompute_coeffs + normalize_coeffs_8bpc for the LANCZOS filter, full-image box
// (same libm, same operation order as Resample.c, so the integer taps are identical)
// ------------------------------------------------------------------------------------------
namespace {
constexpr int kPrecisionBits = 32 - 8 - 2;

inline double sinc_filter(double x) {
  if (x == 0.0) return 1.0;
  x = x * M_PI;
  return std::sin(x) / x;
}
inline double lanczos_filter(double x) {
  if (-3.0 <= x && x < 3.0) return sinc_filter(x) * sinc_filter(x / 3);
  return 0.0;
}
}  // namespace

int ir_host_lanczos_ksize(int in_size, int out_size) {
  double filterscale = (double)((float)in_size - 0.0f) / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 3.0 * filterscale;
  return (int)std::ceil(support) * 2 + 1;
}

void ir_host_lanczos_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk) {
  const double scale = (double)((float)in_size - 0.0f) / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 3.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  double* k = new double[ksize];
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = 0.0 + (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    int x = 0;
    for (; x < xmax; ++x) {
      const double w = lanczos_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (; x < ksize; ++x) k[x] = 0;
    int32_t* row = kk + (int64_t)xx * ksize;
    for (x = 0; x < ksize; ++x)
      row[x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << kPrecisionBits)) : (int)(0.5 + k[x] * (1 << kPrecisionBits));
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
  }
  delete[] k;
}

// ------------------------------------------------------------------------------------------
// device: the two 8-bit passes
// ------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned char clip8(int acc) {
  const int v = acc >> kPrecisionBits;  // arithmetic shift, like Resample.c's clip8 lookup index
  return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// one thread per output byte (x fastest, then channel): neighbouring lanes read neighbouring source
// bytes; the taps of one output column are a broadcast read.
__global__ void __launch_bounds__(256) lanczos_horizontal_kernel(PreprocessKParams p) {
  const ResampleImageK& im = p.img[blockIdx.z];
  const int y = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (y >= im.row_count || j >= p.size * 3) return;
  const int xx = j / 3 + im.crop_left, c = j % 3;
  const int xmin = im.bounds_h[2 * xx], xmax = im.bounds_h[2 * xx + 1];
  const int32_t* __restrict__ k = im.kk_h + (int64_t)xx * im.ksize_h;
  const unsigned char* __restrict__ row = im.src + (int64_t)(im.row_first + y) * im.src_row_bytes + (int64_t)xmin * 3 + c;
  int acc = 1 << (kPrecisionBits - 1);
  for (int t = 0; t < xmax; ++t) acc += (int)row[t * 3] * k[t];
  im.tmp[(int64_t)y * p.size * 3 + j] = clip8(acc);
}

// vertical pass of the crop + ToTensor + Normalize + HWC -> CHW + cast, one thread per output value
template <typename T>
__global__ void __launch_bounds__(256) lanczos_vertical_normalize_kernel(PreprocessKParams p, T* __restrict__ out) {
  const ResampleImageK& im = p.img[blockIdx.z];
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= p.size * 3) return;
  const int yo = blockIdx.y, yy = yo + im.crop_top;
  const int ymin = im.bounds_v[2 * yy] - im.row_first, ymax = im.bounds_v[2 * yy + 1];
  const int32_t* __restrict__ k = im.kk_v + (int64_t)yy * im.ksize_v;
  const unsigned char* __restrict__ col = im.tmp + (int64_t)ymin * p.size * 3 + j;
  int acc = 1 << (kPrecisionBits - 1);
  for (int t = 0; t < ymax; ++t) acc += (int)col[(int64_t)t * p.size * 3] * k[t];
  const float v = (float)clip8(acc);
  // ToTensor: v / 255 (correctly rounded fp32 division); Normalize: (x - 0.5) / 0.5
  const float f = __fsub_rn(__fdiv_rn(v, 255.0f), 0.5f) * 2.0f;
  const int x = j / 3, c = j % 3;
  out[(((int64_t)(p.first_image + blockIdx.z) * 3 + c) * p.size + yo) * p.size + x] = (T)f;
}

// ------------------------------------------------------------------------------------------
// FreeU Fourier filter: one wave per plane, EPL elements per lane (element e*64 + lane)
// ------------------------------------------------------------------------------------------
template <typename T, int EPL>
__global__ void __launch_bounds__(256) freeu_fourier_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t planes,
                                                            int H, int W, int64_t sp_in, int64_t sp_out, int thr,
                                                            float gain /* (scale-1)/(H*W) */) {
  const int lane = threadIdx.x & 63;
  const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (plane >= planes) return;
  const int n = H * W;
  const T* px = x + plane * sp_in;
  float xv[EPL], dl[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int idx = e * 64 + lane;
    xv[e] = idx < n ? (float)px[idx] : 0.f;
    dl[e] = 0.f;
  }
  const float inv_h = 2.0f / (float)H, inv_w = 2.0f / (float)W;  // angles in units of pi
  for (int u = -thr; u < thr; ++u) {
    for (int v = -thr; v < thr; ++v) {
      float re = 0.f, im = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int idx = e * 64 + lane;
        const int r = idx / W, c = idx - r * W;
        const int kr = ((u * r) % H + H) % H, kc = ((v * c) % W + W) % W;
        float sr, cr, sc, cc;
        sincospif((float)kr * inv_h, &sr, &cr);
        sincospif((float)kc * inv_w, &sc, &cc);
        const float co = cr * cc - sr * sc, si = sr * cc + cr * sc;
        re += xv[e] * co;   // X(u,v) = sum x e^{-i theta}
        im -= xv[e] * si;
      }
#pragma unroll
      for (int m = 32; m >= 1; m >>= 1) {
        re += __shfl_xor(re, m, 64);
        im += __shfl_xor(im, m, 64);
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int idx = e * 64 + lane;
        const int r = idx / W, c = idx - r * W;
        const int kr = ((u * r) % H + H) % H, kc = ((v * c) % W + W) % W;
        float sr, cr, sc, cc;
        sincospif((float)kr * inv_h, &sr, &cr);
        sincospif((float)kc * inv_w, &sc, &cc);
        const float co = cr * cc - sr * sc, si = sr * cc + cr * sc;
        dl[e] += re * co - im * si;   // Re(X e^{+i theta})
      }
    }
  }
  T* po = out + plane * sp_out;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int idx = e * 64 + lane;
    if (idx < n) po[idx] = (T)(xv[e] + gain * dl[e]);
  }
}

template <typename T>
hipError_t launch_freeu_t(const void* x, void* out, int64_t planes, int H, int W, int64_t sp_in, int64_t sp_out, int thr,
                          float scale, hipStream_t s) {
  const int n = H * W;
  const float gain = (scale - 1.0f) / (float)n;
  const dim3 g((unsigned)((planes + 3) / 4)), t(256);
  const T* xi = (const T*)x;
  T* xo = (T*)out;
  if (n <= 64) hipLaunchKernelGGL((freeu_fourier_kernel<T, 1>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  else if (n <= 256) hipLaunchKernelGGL((freeu_fourier_kernel<T, 4>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  else if (n <= 1024) hipLaunchKernelGGL((freeu_fourier_kernel<T, 16>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  else hipLaunchKernelGGL((freeu_fourier_kernel<T, 64>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  return hipGetLastError();
}
}  // namespace

hipError_t ir_launch_preprocess(const PreprocessKParams& p, int max_rows, int dtype, void* out, hipStream_t s) {
  const unsigned bx = (unsigned)((p.size * 3 + 255) / 256);
  hipLaunchKernelGGL(lanczos_horizontal_kernel, dim3(bx, (unsigned)max_rows, (unsigned)p.n), dim3(256), 0, s, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const dim3 g(bx, (unsigned)p.size, (unsigned)p.n), t(256);
  if (dtype == 0) hipLaunchKernelGGL((lanczos_vertical_normalize_kernel<_Float16>), g, t, 0, s, p, (_Float16*)out);
  else if (dtype == 1) hipLaunchKernelGGL((lanczos_vertical_normalize_kernel<__bf16>), g, t, 0, s, p, (__bf16*)out);
  else hipLaunchKernelGGL((lanczos_vertical_normalize_kernel<float>), g, t, 0, s, p, (float*)out);
  return hipGetLastError();
}

hipError_t ir_launch_freeu_fourier(const void* x, void* out, int dtype, int64_t planes, int H, int W, int64_t sp_in,
                                   int64_t sp_out, int thr, float scale, hipStream_t s) {
  if (dtype == 0) return launch_freeu_t<_Float16>(x, out, planes, H, W, sp_in, sp_out, thr, scale, s);
  if (dtype == 1) return launch_freeu_t<__bf16>(x, out, planes, H, W, sp_in, sp_out, thr, scale, s);
  return launch_freeu_t<float>(x, out, planes, H, W, sp_in, sp_out, thr, scale, s);
}
