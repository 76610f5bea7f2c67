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

// Horizontal pass.  One workgroup owns kHRows consecutive source rows of one image: it stages the
// byte span under the crop's columns into LDS with aligned dword loads (the span starts at an
// arbitrary byte; all rows of a block share one alignment because they are staged relative to
// their own start), then every thread produces whole pixels for all kHRows rows at once: a tap is
// loaded once (tap-major table: neighbouring lanes, neighbouring ints) and used for kHRows x 3
// multiply-adds; the three bytes of a source pixel come from one two-dword LDS read + v_alignbyte.
template <int kHRows>   // 4 normally; 1 when the span under the crop is too wide for four rows in LDS
__global__ void __launch_bounds__(256) lanczos_horizontal_kernel(PreprocessKParams p, int pitch_dw) {
  extern __shared__ unsigned int lds_rows[];
  const ResampleImageK& im = p.img[blockIdx.y];
  const int row0 = blockIdx.x * kHRows;
  if (row0 >= im.row_count) return;
  const int nrows = min(kHRows, im.row_count - row0);
  const int64_t span0 = (int64_t)im.col_first * 3;           // first byte of the span within a row
  const int span_bytes = im.col_count * 3;
  const unsigned char* img_end = im.src + (int64_t)(im.in_h - 1) * im.src_row_bytes + (int64_t)im.in_w * 3;
  int mis[kHRows];
#pragma unroll
  for (int r = 0; r < kHRows; ++r) {
    const int y = row0 + (r < nrows ? r : 0);                // short blocks recompute row 0 (never stored)
    const unsigned char* g = im.src + (int64_t)(im.row_first + y) * im.src_row_bytes + span0;
    mis[r] = (int)(reinterpret_cast<uintptr_t>(g) & 3u);     // LDS byte i of row r holds g[i - mis[r]]
    const unsigned char* ga = g - mis[r];
    const int ndw = (mis[r] + span_bytes + 3) >> 2;
    unsigned int* dst = lds_rows + r * pitch_dw;
    for (int i = threadIdx.x; i < pitch_dw; i += 256) {      // zero padding past the span: tap groups read it
      const unsigned char* a = ga + 4 * i;
      unsigned int w = 0;
      if (i < ndw) {
        if (a + 4 <= img_end && a >= im.src) {
          w = *reinterpret_cast<const unsigned int*>(a);
        } else {   // first / last dword of the allocation: assemble from the bytes that exist
          for (int b = 0; b < 4; ++b)
            if (a + b >= im.src && a + b < img_end) w |= (unsigned int)a[b] << (8 * b);
        }
      }
      dst[i] = w;
    }
  }
  __syncthreads();
  for (int xl = threadIdx.x; xl < p.size; xl += 256) {
    const int xx = xl + im.crop_left;
    const int xmin = im.bounds_h[2 * xx], xmax = im.bounds_h[2 * xx + 1];
    const int32_t* __restrict__ k = im.kk_h + xx;            // tap-major: k[t * out_w]
    int acc[kHRows][3];
    int off[kHRows];
#pragma unroll
    for (int r = 0; r < kHRows; ++r) {
      acc[r][0] = acc[r][1] = acc[r][2] = 1 << (kPrecisionBits - 1);
      off[r] = (xmin - im.col_first) * 3 + mis[r];           // byte offset of the first tap's pixel in row r
    }
    // taps in groups of four: 4 pixels = 12 bytes = 3 dwords, so the byte phase of a (thread, row)
    // never changes and every byte position inside the realigned window is a compile-time constant.
    // The tap table is zero beyond xmax (Resample.c zero-fills it), the rows are zero-padded in LDS.
    for (int t = 0; t < xmax; t += 4) {
      int kt[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) kt[i] = t + i < im.ksize_h ? k[(int64_t)(t + i) * im.out_w] : 0;
#pragma unroll
      for (int r = 0; r < kHRows; ++r) {
        const int o = off[r] + 3 * t;
        const unsigned int* q = lds_rows + r * pitch_dw + (o >> 2);
        const unsigned int d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
        const unsigned int m = (unsigned)(o & 3);
        const unsigned int e[3] = {__builtin_amdgcn_alignbyte(d1, d0, m), __builtin_amdgcn_alignbyte(d2, d1, m),
                                   __builtin_amdgcn_alignbyte(d3, d2, m)};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int byte = 3 * i + c;
            acc[r][c] += (int)((e[byte >> 2] >> (8 * (byte & 3))) & 255u) * kt[i];
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kHRows; ++r) {
      if (r < nrows) {
        unsigned char* o = im.tmp + (int64_t)(row0 + r) * p.tmp_pitch + xl * 3;
        o[0] = clip8(acc[r][0]); o[1] = clip8(acc[r][1]); o[2] = clip8(acc[r][2]);
      }
    }
  }
}

// Vertical pass of the crop + ToTensor + Normalize + HWC -> CHW + cast.  One thread per dword of
// kVRows consecutive output rows (4 consecutive bytes of the interleaved RGB stream): neighbouring
// output rows share most of their source rows, so the union is walked once - one coalesced dword
// load per source row, the taps are uniform over the workgroup (scalar loads).
constexpr int kVRows = 4;
template <typename T>
__global__ void __launch_bounds__(256) lanczos_vertical_normalize_kernel(PreprocessKParams p, T* __restrict__ out) {
  const ResampleImageK& im = p.img[blockIdx.z];
  const int yo0 = blockIdx.y * kVRows;
  // the taps of this block's rows go to LDS once (a dependent scalar load per tap in the loop below
  // would serialise on memory latency); aligned to the union's first source row so the inner loop
  // indexes them with the loop counter: taps[r][y - lo], zero outside the row's own window
  extern __shared__ int lds_taps[];
  // ToTensor + Normalize of the 256 possible bytes, once per workgroup: v / 255 (correctly rounded
  // fp32 division), (x - 0.5) / 0.5 - the per-value division is what the epilogue would otherwise pay
  __shared__ T lut[256];
  lut[threadIdx.x] = (T)(__fsub_rn(__fdiv_rn((float)threadIdx.x, 255.0f), 0.5f) * 2.0f);
  int lo = 0x7fffffff, hi = 0;
#pragma unroll
  for (int r = 0; r < kVRows; ++r) {
    const int yy = min(yo0 + r, p.size - 1) + im.crop_top;
    const int y0 = im.bounds_v[2 * yy] - im.row_first;
    lo = min(lo, y0);
    hi = max(hi, y0 + im.bounds_v[2 * yy + 1]);
  }
  const int span = hi - lo;                       // <= ksize_v + (kVRows - 1) * ceil(scale) <= 2 * ksize_v + kVRows
  for (int i = threadIdx.x; i < kVRows * span; i += 256) {
    const int r = i / span, y = lo + (i - r * span);
    const int yy = min(yo0 + r, p.size - 1) + im.crop_top;
    const int t = y - (im.bounds_v[2 * yy] - im.row_first);
    int v = 0;
    if (yo0 + r < p.size && t >= 0 && t < im.bounds_v[2 * yy + 1]) v = im.kk_v[(int64_t)yy * im.ksize_v + t];
    lds_taps[i] = v;
  }
  __syncthreads();
  const int j4 = blockIdx.x * 256 + threadIdx.x;
  const int nbytes = p.size * 3;
  if (j4 * 4 >= nbytes) return;
  const int pitch4 = p.tmp_pitch >> 2;
  const unsigned int* __restrict__ col = reinterpret_cast<const unsigned int*>(im.tmp) + (int64_t)lo * pitch4 + j4;
  int acc[kVRows][4];
#pragma unroll
  for (int r = 0; r < kVRows; ++r)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[r][b] = 1 << (kPrecisionBits - 1);
#pragma unroll 4
  for (int y = 0; y < span; ++y) {
    const unsigned int w = col[(int64_t)y * pitch4];
#pragma unroll
    for (int r = 0; r < kVRows; ++r) {
      const int kt = lds_taps[r * span + y];      // broadcast read
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[r][b] += (int)((w >> (8 * b)) & 255u) * kt;
    }
  }
#pragma unroll
  for (int r = 0; r < kVRows; ++r) {
    const int yo = yo0 + r;
    if (yo >= p.size) break;
    T* po = out + (int64_t)(p.first_image + blockIdx.z) * 3 * p.size * p.size + (int64_t)yo * p.size;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int j = j4 * 4 + b;
      if (j < nbytes) {
        const int x = j / 3, c = j - 3 * x;
        po[(int64_t)c * p.size * p.size + x] = lut[clip8(acc[r][b])];
      }
    }
  }
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

// threshold == 1 (the only value the reference uses, block.py:3514,3518): the four bins are
// (0,0), (-1,0), (0,-1), (-1,-1), i.e. seven real sums against cos/sin of 2 pi r/H, 2 pi c/W and
// their sum.  The H + W phase values are computed once per workgroup into LDS; an element costs
// four LDS reads and a dozen FMAs in each of the two sweeps (sums, then apply).
template <typename T, int EPL>
__global__ void __launch_bounds__(256) freeu_fourier_t1_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t planes,
                                                               int H, int W, int64_t sp_in, int64_t sp_out, float gain) {
  extern __shared__ float2 lds_phase[];   // [0, H): (cos, sin)(2 pi r / H); [H, H + W): same for columns
  for (int i = threadIdx.x; i < H + W; i += 256) {
    float sn, cs;
    if (i < H) sincospif((float)i * (2.0f / (float)H), &sn, &cs);
    else sincospif((float)(i - H) * (2.0f / (float)W), &sn, &cs);
    lds_phase[i] = make_float2(cs, sn);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int64_t plane = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (plane >= planes) return;
  const int n = H * W;
  const T* px = x + plane * sp_in;
  const float rw = 1.0f / (float)W;
  float xv[EPL];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f;
  auto phases = [&](int idx, float2& pr, float2& pc) {
    int r = (int)(((float)idx + 0.5f) * rw);   // idx / W without an integer divide, fixed up below
    int col = idx - r * W;
    if (col < 0) { col += W; --r; }
    if (col >= W) { col -= W; ++r; }
    pr = lds_phase[r];
    pc = lds_phase[H + col];
  };
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int idx = e * 64 + lane;
    const float v = idx < n ? (float)px[idx] : 0.f;
    xv[e] = v;
    float2 pr, pc;
    phases(idx < n ? idx : 0, pr, pc);
    s0 += v;
    s1 += v * pr.x; s2 += v * pr.y;
    s3 += v * pc.x; s4 += v * pc.y;
    s5 += v * (pr.x * pc.x - pr.y * pc.y); s6 += v * (pr.y * pc.x + pr.x * pc.y);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    s0 += __shfl_xor(s0, m, 64); s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64);
    s3 += __shfl_xor(s3, m, 64); s4 += __shfl_xor(s4, m, 64); s5 += __shfl_xor(s5, m, 64);
    s6 += __shfl_xor(s6, m, 64);
  }
  T* po = out + plane * sp_out;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const int idx = e * 64 + lane;
    if (idx < n) {
      float2 pr, pc;
      phases(idx, pr, pc);
      const float delta = s0 + s1 * pr.x + s2 * pr.y + s3 * pc.x + s4 * pc.y +
                          s5 * (pr.x * pc.x - pr.y * pc.y) + s6 * (pr.y * pc.x + pr.x * pc.y);
      po[idx] = (T)(xv[e] + gain * delta);
    }
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
  if (thr == 1) {
    const size_t lds = (size_t)(H + W) * sizeof(float2);
    if (n <= 64) hipLaunchKernelGGL((freeu_fourier_t1_kernel<T, 1>), g, t, lds, s, xi, xo, planes, H, W, sp_in, sp_out, gain);
    else if (n <= 256) hipLaunchKernelGGL((freeu_fourier_t1_kernel<T, 4>), g, t, lds, s, xi, xo, planes, H, W, sp_in, sp_out, gain);
    else if (n <= 1024) hipLaunchKernelGGL((freeu_fourier_t1_kernel<T, 16>), g, t, lds, s, xi, xo, planes, H, W, sp_in, sp_out, gain);
    else hipLaunchKernelGGL((freeu_fourier_t1_kernel<T, 64>), g, t, lds, s, xi, xo, planes, H, W, sp_in, sp_out, gain);
    return hipGetLastError();
  }
  if (n <= 64) hipLaunchKernelGGL((freeu_fourier_kernel<T, 1>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  else if (n <= 256) hipLaunchKernelGGL((freeu_fourier_kernel<T, 4>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  else if (n <= 1024) hipLaunchKernelGGL((freeu_fourier_kernel<T, 16>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  else hipLaunchKernelGGL((freeu_fourier_kernel<T, 64>), g, t, 0, s, xi, xo, planes, H, W, sp_in, sp_out, thr, gain);
  return hipGetLastError();
}
}  // namespace

hipError_t ir_launch_preprocess(const PreprocessKParams& p, int max_rows, int max_span_bytes, int max_ksize_v, int dtype,
                                void* out, hipStream_t s) {
  const int pitch_dw = (max_span_bytes + 3 + 3) / 4 + 4 + 3;   // + one tap group + the realignment window
  if ((size_t)pitch_dw * 4 * 4 <= 60 * 1024) {
    hipLaunchKernelGGL(lanczos_horizontal_kernel<4>, dim3((unsigned)((max_rows + 3) / 4), (unsigned)p.n), dim3(256),
                       (size_t)pitch_dw * 4 * 4, s, p, pitch_dw);
  } else {   // very wide sources: one row per workgroup keeps the staged span within LDS
    hipLaunchKernelGGL(lanczos_horizontal_kernel<1>, dim3((unsigned)max_rows, (unsigned)p.n), dim3(256),
                       (size_t)pitch_dw * 4, s, p, pitch_dw);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const unsigned bx = (unsigned)(((p.size * 3 + 3) / 4 + 255) / 256);
  const dim3 g(bx, (unsigned)((p.size + kVRows - 1) / kVRows), (unsigned)p.n), t(256);
  const size_t vlds = (size_t)kVRows * (2 * max_ksize_v + kVRows) * sizeof(int);
  if (dtype == 0) hipLaunchKernelGGL((lanczos_vertical_normalize_kernel<_Float16>), g, t, vlds, s, p, (_Float16*)out);
  else if (dtype == 1) hipLaunchKernelGGL((lanczos_vertical_normalize_kernel<__bf16>), g, t, vlds, s, p, (__bf16*)out);
  else hipLaunchKernelGGL((lanczos_vertical_normalize_kernel<float>), g, t, vlds, s, p, (float*)out);
  return hipGetLastError();
}

hipError_t ir_launch_freeu_fourier(const void* x, void* out, int dtype, int64_t planes, int H, int W, int64_t sp_in,
                                   int64_t sp_out, int thr, float scale, hipStream_t s) {
  if (dtype == 0) return launch_freeu_t<_Float16>(x, out, planes, H, W, sp_in, sp_out, thr, scale, s);
  if (dtype == 1) return launch_freeu_t<__bf16>(x, out, planes, H, W, sp_in, sp_out, thr, scale, s);
  return launch_freeu_t<float>(x, out, planes, H, W, sp_in, sp_out, thr, scale, s);
}
