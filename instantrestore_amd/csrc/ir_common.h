// ir_common.h - shared device/host helpers for the gfx950 (CDNA4) kernels of instantrestore_amd.
// Written for MI355X only: 64-wide wavefronts, v_mfma_f32_32x32x16_{bf16,f16}, ds_read_b64_tr_b16.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define IR_LDS __attribute__((address_space(3)))

// Element-type traits: the 2-byte storage type T selects the MFMA flavour.
template <typename T>
struct ElemTraits;

template <>
struct ElemTraits<__bf16> {
  using v8 = bf16x8;
  using v4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <>
struct ElemTraits<_Float16> {
  using v8 = f16x8;
  using v4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// ds_read_b64_tr_b16: within each 16-lane group, lane i receives element (i & 3) of the 8 bytes
// addressed by lanes (i >> 2) + 4*j, j = 0..3 (a 4x16 -> 16x4 transpose of 16-bit elements).
static __device__ __forceinline__ s16x4 lds_read_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((IR_LDS s16x4*)(p));
}

template <typename V8>
static __device__ __forceinline__ V8 join_tr(s16x4 lo, s16x4 hi) {
  s16x8 j = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(V8, j);
}

// XCD-aware block remap (8 XCDs, block b is dispatched to XCD b % 8): gives every XCD a
// CONTIGUOUS range of logical work items so blocks that share K/V share an L2.  Bijective for any
// grid size.  Speed only - correctness never depends on placement.
static __device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + slot;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// Lanes of ONE wave exchanging data through LDS (the GEMM epilogues: 16-bit rows written in the MFMA layout, read back as
// whole 16-byte pieces of other lanes' rows) rely on the hardware executing a wave's LDS operations in order - and on the
// COMPILER keeping them in order, which the language does not promise: to hipcc every lane is a thread of its own, and a
// thread that executed no store between two loads of one address may have the second load forwarded from the first.  Round
// 4: an unrelated edit of the 256 x 256 epilogue (whose staging writes sit under a lane-divergent `if`) made hipcc do
// exactly that - rows 16-31 of every 32-row block came out as copies of rows 0-15, deterministically
// (tests/test_gpu_linear.py).  ir_wave_lds_fence() goes between such writes and reads, both ways: a compiler barrier over
// memory (no instruction; the hardware order is already there).  The staging reads also go through may_alias types.
static __device__ __forceinline__ void ir_wave_lds_fence() { asm volatile("" ::: "memory"); }
typedef u32x4 __attribute__((may_alias)) u32x4_alias;
typedef f32x4 __attribute__((may_alias)) f32x4_alias;

// 3-input max in one VALU op. Plain fmaxf() chains make hipcc canonicalise every MFMA output
// first (a v_max_f32 x,x,x per element); the asm form takes the raw registers.
static __device__ __forceinline__ float max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// LDS-DMA: 16 bytes per lane straight from a buffer resource into LDS at (wave-uniform base +
// lane*16).  Kept in a __device__ helper: the host pass must not see the address-space cast.
static __device__ __forceinline__ void buffer_load_lds16(__amdgpu_buffer_rsrc_t rsrc, unsigned char* lds_base,
                                                         unsigned voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (IR_LDS void*)lds_base, 16, voffset, 0, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;

// Raw buffer descriptor words (base, stride 0, num_records, flags) for hand-issued buffer ops.
static __device__ __forceinline__ i32x4 make_rsrc_words(const void* base, unsigned num_records) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = (int)(unsigned)a;
  r[1] = (int)((unsigned)(a >> 32) & 0xffffu);
  r[2] = (int)num_records;
  r[3] = 0x00020000;
  return r;
}

// LDS-DMA issued from inline asm: invisible to hipcc's waitcnt bookkeeping, so the transfer stays
// in flight across LDS reads until OUR s_waitcnt vmcnt(0) (placed before the step's barrier).
// M0 (the LDS destination base) is written in the same statement that consumes it.
static __device__ __forceinline__ void buffer_load_lds16_async(i32x4 rsrc, unsigned char* lds_base, unsigned voffset) {
  const unsigned lds_addr = (unsigned)(unsigned long long)(IR_LDS unsigned char*)lds_base;
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_addr), "v"(voffset), "s"(rsrc)
               : "memory");
}

// 16-byte store through a buffer resource, issued from asm like the transfers above: byte address = base + voffset (per
// lane; what the hardware range-checks against num_records: lanes past it are dropped) + soffset (wave-uniform, SGPR).
// A kernel that sends every store of a wave through ONE per-lane offset keeps one register where per-store 64-bit
// addresses were sixteen (linear_skinny.hip).  Counts in vmcnt like any store; hipcc's own waits only get stricter.
// Cache policy of the projection GEMMs' OUTPUT stores: non-temporal since round 5 (-DIR_LIN_STORE_NT=0 rebuilds the plain form).
// Y is written once by the GEMM and is larger than the L2s for every big shape; without the hint its lines push the operand
// panels out.  Same-box A/B over the twelve step shapes (round-5 A/B driver, three alternations; profiles/r5_gemm_probe_final.txt): 1.79 -> 1.71 ms per step of
// GEMM time in isolation, 131072 x 960 x 320 with fp32 activations 121 -> 100 us; the two-stream step itself is within its noise
// (7.06 vs 7.02 ms): under the power cap and beside the other stream's attention the GEMMs are not what the step waits for.
#ifndef IR_LIN_STORE_NT
#define IR_LIN_STORE_NT 1
#endif
static __device__ __forceinline__ void buffer_store16_async(i32x4 rsrc, u32x4 v, unsigned voffset, unsigned soffset) {
#if IR_LIN_STORE_NT
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt" : : "v"(v), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
#else
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" : : "v"(v), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
#endif
}
static __device__ __forceinline__ void ir_store_y(u32x4* p, u32x4 v) {
#if IR_LIN_STORE_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

static __device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Opt-in to more than 64 KiB of dynamic LDS (hipFuncAttributeMaxDynamicSharedMemorySize): the attribute belongs to
// (function, CURRENT device) and setting it is idempotent.  Each launcher instantiation keeps one IrOncePerDevice (a static
// of the template function): an atomic flag per device, no lock - a race between threads only repeats an idempotent call,
// and a device index beyond the table makes the call on every launch.  This is the only state of the library that outlives
// a call (round 6: atomics; rounds 3-5 used plain bools here).
struct IrOncePerDevice {
  std::atomic<unsigned char> done[64];
};
static inline hipError_t ir_opt_in_dynamic_lds(IrOncePerDevice& once, const void* fn, size_t bytes, int* device = nullptr) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  if (device != nullptr) *device = dev;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && once.done[dev].load(std::memory_order_acquire) != 0) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess && tracked) once.done[dev].store(1, std::memory_order_release);
  return e;
}
