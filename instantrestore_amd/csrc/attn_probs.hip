// attn_probs.hip - materialise attention_probs (B,H,Lq,Lkv) for the dump path (gfx950).
//
// face_replace/models/attn_processors.py:258-261 keeps the probability matrix when
// save_self_attentions is set (read by test.py:108, gradio_demo.py:118, coach.py:312).  The fused
// forward never forms it; this kernel recomputes  P = exp(scale*QK^T - LSE)  from the LSE the fused
// forward emitted and streams it out.  HBM-write bound by construction (H*L*Lkv*2 bytes); the
// QK^T recompute rides on MFMA straight from global/L2 (no LDS: each K row is used once per wave).
#include "ir_common.h"
#include "ir_kernels.h"

namespace {

constexpr int PW = 4;  // waves per workgroup, 32 query rows each

template <typename T>
__global__ void __launch_bounds__(PW * 64) attn_probs_kernel(const AttnKParams p) {
  using Tr = ElemTraits<T>;
  using v8 = typename Tr::v8;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hi = lane >> 5, lq = lane & 31;
  const int nqb = (p.Lq + PW * 32 - 1) / (PW * 32);
  const int bh = blockIdx.x / nqb;
  const int qb = blockIdx.x - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * (PW * 32) + wid * 32;
  if (q0 >= p.Lq) return;

  // A operand: Q[row lq][d = 16ks + 8hi ..]
  const int qr = (q0 + lq < p.Lq) ? q0 + lq : p.Lq - 1;
  const T* qp = (const T*)p.q + (int64_t)b * p.q_sb + (int64_t)qr * p.q_sl + (int64_t)h * p.q_sh + hi * 8;
  v8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const v8*)(qp + ks * 16);

  // output rows of this lane: q0 + crow(r,hi); their LSE in the exp2 domain
  const float LOG2E = 1.4426950408889634f;
  float lse2[16];
  int64_t orow[16];
  bool rok[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
    rok[r] = q < p.Lq;
    const int qc = rok[r] ? q : p.Lq - 1;
    lse2[r] = p.lse[((int64_t)b * p.H + h) * p.Lq + qc] * LOG2E;
    orow[r] = (((int64_t)b * p.H + h) * p.Lq + qc) * (int64_t)p.lkv;
  }
  T* probs = (T*)p.probs;

  const int nseg = p.include_self + p.N;
  int col0 = 0;
  for (int s = 0; s < nseg; ++s) {
    const T* kb;
    int64_t ksl;
    int len;
    if (p.include_self && s == 0) {
      kb = (const T*)p.k_self + (int64_t)b * p.ks_sb + (int64_t)h * p.ks_sh; ksl = p.ks_sl; len = p.Ls;
    } else {
      const int n = s - p.include_self;
      kb = (const T*)p.k_ref + (int64_t)b * p.kr_sb + (int64_t)n * p.kr_sn + (int64_t)h * p.kr_sh; ksl = p.kr_sl; len = p.Lr;
    }
    for (int j0 = 0; j0 < len; j0 += 32) {
      const int key = j0 + lq;
      const bool kok = key < len;
      const T* kp = kb + (int64_t)(kok ? key : len - 1) * ksl + hi * 8;
      f32x16 sc;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sc = Tr::mfma(qf[ks], *(const v8*)(kp + ks * 16), sc);
      // sc[r] = <Q[q0+crow(r,hi)], K[key]> ; column = lane & 31 = key
      if (kok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (rok[r]) probs[orow[r] + col0 + key] = (T)fast_exp2(__builtin_fmaf(sc[r], p.scale_log2, -lse2[r]));
        }
      }
    }
    col0 += len;
  }
}

}  // namespace

hipError_t ir_launch_attn_probs(const AttnKParams& p, int dtype, hipStream_t s) {
  const int nqb = (p.Lq + PW * 32 - 1) / (PW * 32);
  const dim3 grid(p.B * p.H * nqb);
  if (dtype == 1) hipLaunchKernelGGL((attn_probs_kernel<__bf16>), grid, dim3(PW * 64), 0, s, p);
  else hipLaunchKernelGGL((attn_probs_kernel<_Float16>), grid, dim3(PW * 64), 0, s, p);
  return hipGetLastError();
}
