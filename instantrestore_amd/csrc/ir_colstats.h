// ir_colstats.h - token statistics of freshly written projection outputs, as the TAIL of the GEMM that wrote them.
//
// adain() (face_replace/models/attn_processors.py:9-10, call site :244-245) needs mean and unbiased std over the token
// axis of every V matrix.  Rounds 1-3 took them in a separate pass over V (adain.hip: adain_partial_kernel); since round 4
// every WAVE of the q/k/v projection that has just stored its 64 rows of the V third re-reads them - its own stores, still
// in this XCD's L2 - and leaves one partial (mean[64], M2[64]) per (64-row block, head) in a workspace that
// adain.hip's adain_affine_partials_kernel / token_stats_partials_kernel merge.  The statistics are those of the ROUNDED
// outputs, i.e. of the 16-bit values the attention kernel reads, exactly like the standalone pass; no HBM read, no launch,
// no LDS and no barrier: a wave only ever reads back rows it stored itself.
//
// Arithmetic: sums of (x - K) and (x - K)^2 with K = the block's first row (one shift per column, shared by the eight
// lanes that walk a column group, so their sums add up directly), butterfly over the row slots; mean = K + s1 / 64,
// M2 = s2 - s1^2 / 64.  The first version of this file merged per-thread Chan partials through LDS behind two barriers
// per head, one head at a time: +0.07 ms on the step instead of -0.2 (profiles/r4_ab_fused_stats.txt).
#pragma once
#include "ir_common.h"

constexpr int kStatsRows = 64;   // rows per statistics block = rows of Y a wave owns in every projection kernel

// Accumulator of one wave for one (64-row block, head): lane l walks column group cs = l & 7 (eight columns) over the rows
// rs + 8 j, rs = l >> 3.  Every projection kernel's epilogue hands its lanes the finished 16-bit rows in exactly this
// shape - eight columns of row 8 j + (l >> 3) - on their way to the whole-line stores, so the LDS-tiled kernels feed the
// statistics from those registers (no load at all); the X-stationary kernels, which have no registers to spare in their
// loop, re-read their own stores behind it (ir_wave_col_stats).
struct ColStatsAcc {
  float K[8], s1[8], s2[8];
};

// first visit (the one in which lanes 0-7 hold row 0 of the block): takes the shift of each column from that row
template <typename T>
static __device__ __forceinline__ void ir_stats_first(ColStatsAcc& a, u32x4 bits) {
  using v8 = typename ElemTraits<T>::v8;
  const f32x8 f = __builtin_convertvector(__builtin_bit_cast(v8, bits), f32x8);
  const int src = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 7;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a.K[i] = __shfl(f[i], src);
    const float d = f[i] - a.K[i];
    a.s1[i] = d;
    a.s2[i] = d * d;
  }
}

template <typename T>
static __device__ __forceinline__ void ir_stats_add(ColStatsAcc& a, u32x4 bits) {
  using v8 = typename ElemTraits<T>::v8;
  const f32x8 f = __builtin_convertvector(__builtin_bit_cast(v8, bits), f32x8);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float d = f[i] - a.K[i];
    a.s1[i] += d;
    a.s2[i] = __builtin_fmaf(d, d, a.s2[i]);
  }
}

// butterfly over the row slots (lane bits 3..5: every lane ends with the block's totals, in one fixed order), then lanes
// 0-7 store mean[8] and M2[8] of their column group.  wsp: &ws[(block * nheads + head) * 128]
static __device__ __forceinline__ void ir_stats_finish(ColStatsAcc& a, float* __restrict__ wsp) {
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a.s1[i] += __shfl_xor(a.s1[i], m);
      a.s2[i] += __shfl_xor(a.s2[i], m);
    }
  }
  if ((lane >> 3) == 0) {
    float* o = wsp + (lane & 7) * 8;
    f32x4 m0, m1, q0, q1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m0[i] = a.K[i] + a.s1[i] * (1.0f / kStatsRows);
      m1[i] = a.K[4 + i] + a.s1[4 + i] * (1.0f / kStatsRows);
      q0[i] = fmaxf(a.s2[i] - a.s1[i] * a.s1[i] * (1.0f / kStatsRows), 0.f);
      q1[i] = fmaxf(a.s2[4 + i] - a.s1[4 + i] * a.s1[4 + i] * (1.0f / kStatsRows), 0.f);
    }
    *(f32x4*)o = m0;
    *(f32x4*)(o + 4) = m1;
    *(f32x4*)(o + 64) = q0;
    *(f32x4*)(o + 68) = q1;
  }
}

// One wave: partials of heads [head_lo, head_hi) of the statistics range (st_col0, st_cols) over rows [row0, row0 + 64) of
// y, which THIS wave stored (the caller has waited for those stores).  ws layout: [row0 / 64][st_cols / 64][128].  The loads
// of up to HB heads are in flight together: one head at a time was one L2 round trip per head, +10 us on the K = 320 GEMM.
template <typename T, int HB>
static __device__ __forceinline__ void ir_wave_col_stats(const T* __restrict__ y, int64_t y_ld, int row0, int head_lo, int head_hi,
                                                         float* __restrict__ st_ws, int st_col0, int st_cols) {
  // the lane index is formed HERE (v_mbcnt) behind a statement the compiler cannot move: derived from the kernels' own
  // `lane` it was computed in their prologues and carried - spilled - across their main loops (tools/check_resources.py)
  int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane));
  const int cs = lane & 7, rs = lane >> 3;
  const int nheads = st_cols >> 6;
  const T* const base = y + (int64_t)row0 * y_ld + st_col0 + cs * 8;
  for (int h0 = head_lo; h0 < head_hi; h0 += HB) {
    u32x4 xs[HB][8];
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
      const int head = (h0 + hb) < head_hi ? (h0 + hb) : (head_hi - 1);    // past the range: the last head again (not stored)
#pragma unroll
      // These loads re-read what this workgroup's own waves just stored.  WHAT GUARANTEES FRESH DATA is the C ABI's alignment
      // rule for a statistics call (c_abi.hip: y_ld * 2 % 128 == 0 and the statistics columns on 128-B line boundaries), under
      // which no 128-B line of y is shared between the column ranges of two workgroups: a line this wave reads was written by
      // this workgroup only, after the barrier, and was not in this CU's vector L1 before (stores write through and invalidate
      // the line).  The `nt` hint is a STREAMING hint (do not keep the line), not a coherence control: on gfx942 / gfx950 a stale
      // line already in L1 could still hit under it - bypassing L1 would take sc0 sc1 on the load or a buffer_inv sc1 before it.
      // It is kept for its cache effect only (the lines are read once).
      for (int j = 0; j < 8; ++j) xs[hb][j] = __builtin_nontemporal_load((const u32x4*)(base + head * 64 + (int64_t)(rs + 8 * j) * y_ld));
    }
#pragma unroll
    for (int hb = 0; hb < HB; ++hb) {
      if (h0 + hb < head_hi) {
        ColStatsAcc a;
        ir_stats_first<T>(a, xs[hb][0]);
#pragma unroll
        for (int j = 1; j < 8; ++j) ir_stats_add<T>(a, xs[hb][j]);
        ir_stats_finish(a, st_ws + ((int64_t)(row0 / kStatsRows) * nheads + h0 + hb) * 128);
      }
    }
  }
}

// One wave: partial of ONE head from a 64-row x 64-column block of finished 16-bit outputs that sits in LDS (row pitch `pitch`
// bytes), e.g. the staging tile of the X-stationary kernels right after a chunk pair has left.  Same sums in the same order as
// ir_stats_first / _add / _finish (shift = row 0, rows rs + 8 j, butterfly over lane bits 3..5) - the same bits - but built
// for a kernel with ~35 registers to spare: four columns at a time, rows in groups of 3-4 (12 accumulators + ~16 temporaries).
template <typename T>
static __device__ __forceinline__ void ir_lds_block_stats(const unsigned char* tb, int pitch, float* __restrict__ wsp) {
  using v4 = typename ElemTraits<T>::v4;
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2), may_alias));
  int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane));
  const int cs = lane & 7, rs = lane >> 3;
#pragma nounroll
  for (int half = 0; half < 2; ++half) {
    const unsigned char* src = tb + rs * pitch + cs * 16 + half * 8;
    float K[4], s1[4], s2[4];
    {
      const f32x4 f = __builtin_convertvector(__builtin_bit_cast(v4, *(const u32x2_*)src), f32x4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        K[i] = __shfl(f[i], cs);
        const float d = f[i] - K[i];
        s1[i] = d;
        s2[i] = d * d;
      }
    }
    // rows 1-3, then 4-7: the loads of a group in flight together (one LDS round trip per group, not per row); same order of sums
#pragma unroll
    for (int j0 = 1; j0 < 8; j0 += (j0 == 1 ? 3 : 4)) {
      const int nj = j0 == 1 ? 3 : 4;
      u32x2_ raw[4];
#pragma unroll
      for (int jj = 0; jj < nj; ++jj) raw[jj] = *(const u32x2_*)(src + 8 * (j0 + jj) * pitch);
#pragma unroll
      for (int jj = 0; jj < nj; ++jj) {
        const f32x4 f = __builtin_convertvector(__builtin_bit_cast(v4, raw[jj]), f32x4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d = f[i] - K[i];
          s1[i] += d;
          s2[i] = __builtin_fmaf(d, d, s2[i]);
        }
      }
    }
#pragma unroll
    for (int m = 8; m < 64; m <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s1[i] += __shfl_xor(s1[i], m);
        s2[i] += __shfl_xor(s2[i], m);
      }
    }
    if (rs == 0) {
      f32x4 mean, m2;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mean[i] = K[i] + s1[i] * (1.0f / kStatsRows);
        m2[i] = fmaxf(s2[i] - s1[i] * s1[i] * (1.0f / kStatsRows), 0.f);
      }
      float* o = wsp + cs * 8 + half * 4;
      *(f32x4*)o = mean;
      *(f32x4*)(o + 64) = m2;
    }
  }
}

// heads [lo, hi) of the statistics range that lie inside columns [n0, n1) (all multiples of 64)
static __device__ __forceinline__ void ir_stats_heads(int st_col0, int st_cols, int n0, int n1, int& lo, int& hi) {
  const int a = n0 - st_col0, b = n1 - st_col0, nh = st_cols >> 6;
  lo = a > 0 ? (a >> 6) : 0;
  hi = b > 0 ? ((b >> 6) < nh ? (b >> 6) : nh) : 0;
}
