// Shared device/host helpers of libganconv (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "ganconv.h"

namespace gc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;   // 4 wavefronts of 64
constexpr int kBK = 64;         // reduction-tile depth (channels) of the gather kernels, in elements
constexpr int kMaxTaps = 64;
constexpr int kMaxClasses = 16;

// fp32 x4 -> bf16 x4 (round to nearest even; v_cvt_pk_bf16_f32), returned as two packed dwords
__device__ __forceinline__ u32x2 cvt_bf16x4(f32x4 v) {
  bf16x4 b = __builtin_convertvector(v, bf16x4);
  return __builtin_bit_cast(u32x2, b);
}
__device__ __forceinline__ f32x4 widen_bf16x4(u32x2 v) {
  f32x4 r;
  r[0] = __builtin_bit_cast(float, v[0] << 16);
  r[1] = __builtin_bit_cast(float, v[0] & 0xffff0000u);
  r[2] = __builtin_bit_cast(float, v[1] << 16);
  r[3] = __builtin_bit_cast(float, v[1] & 0xffff0000u);
  return r;
}
__device__ __forceinline__ float bf16_bits_to_float(uint16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
__device__ __forceinline__ uint16_t float_to_bf16_bits(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(uint16_t, b);
}

// 4 consecutive channels of an fp32 or bf16 tensor as fp32 (element index idx, a multiple of 4)
template <typename T>
__device__ __forceinline__ f32x4 load4(const void* base, size_t idx) {
  if (sizeof(T) == 4) return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  return widen_bf16x4(*reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(base) + idx));
}
template <typename T>
__device__ __forceinline__ void store4(void* base, size_t idx, f32x4 v) {
  if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + idx) = v;
  else *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(base) + idx) = cvt_bf16x4(v);
}

// Unsigned division by a launch-invariant divisor (Granlund-Montgomery round-up form; shift path for powers of two).
struct FastDiv {
  uint32_t d, mul, shift, pow2;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{d, 0, 0, 0};
  if (d == 0) d = 1;
  if ((d & (d - 1)) == 0) {
    f.pow2 = 1;
    while ((1u << f.shift) < d) f.shift++;
    return f;
  }
  uint32_t l = 0;
  while ((1ull << l) < d) l++;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.shift = l - 1;
  return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  if (f.pow2) return n >> f.shift;
  uint32_t q = __umulhi(n, f.mul);
  return (((n - q) >> 1) + q) >> f.shift;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case GC_ACT_RELU: return v > 0.f ? v : 0.f;
    case GC_ACT_LRELU: return v > 0.f ? v : v * slope;
    case GC_ACT_TANH: return tanhf(v);
    case GC_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    default: return v;
  }
}

// Blocks are dealt round-robin over the 8 XCDs (private L2 each): give every XCD one contiguous run of logical
// tile ids so that tiles sharing an activation panel meet in one L2.  Bijective for any grid size.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u, xcd = bid & 7u, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Zero-fill as an ordinary kernel (a memset node recorded into a HIP graph was seen not to take effect on replay).
bool patch_kernel_allowed();     // process-wide switch (gc_set_patch_kernel)
bool split_reduction_allowed();   // process-wide switch (gc_set_split_reduction)
int zero_fill(void* ptr, size_t bytes, hipStream_t stream);   // bytes a multiple of 4, ptr 4-byte aligned

// ---------------------------------------------------------------------------------------------------------
// Host-side bookkeeping shared by the translation units.
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that needs more than 64 KiB, once per (kernel, device) with
// the calling thread's current device; callable from any host thread.
int ensure_dynamic_smem(const void* kernel, int bytes);
// Compute units of the current device (cached per device).
int device_cu_count();

struct ProfScope {  // records a pair of events around a launch when profiling is on
  ProfScope(const char* name, double flops, hipStream_t s);
  ~ProfScope();
  int slot;
  hipStream_t stream;
};

}  // namespace gc
