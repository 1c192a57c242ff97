// Channel concatenation of two NHWC tensors fused with the (Leaky)ReLU that follows it, and the adjoint:
//
//   forward    y[p, 0 .. Ca)       = act(a[p, :]),   y[p, Ca .. Ca + Cb) = act(b[p, :]),   act(v) = v > 0 ? v : slope * v
//   backward   ga[p, :] = g[p, 0 .. Ca) * m,  gb[p, :] = g[p, Ca .. Ca + Cb) * m,           m = y > 0 ? 1 : slope
//
// The skip connections of the Pix2Pix U-Net (Pix2Pix/models.py:128-145: `up(torch.cat((out, skip), dim=1))` with the
// block's ReLU in front of its transposed convolution, models.py:30-32) run cat, ReLU, ReLU's backward and the two
// slices of cat's backward as five elementwise launches over the doubled tensor; here one launch per direction.
// slope = 1 is a plain concatenation / split.  Memory-bound: 16-byte accesses, consecutive lanes along the output row.
#include <initializer_list>

#include "common.h"

namespace gc {
namespace {

template <typename T>
struct Vec16 {   // 16 bytes of storage: 4 floats or 8 bf16
  static constexpr int kElems = 16 / sizeof(T);
};

__device__ __forceinline__ float act1(float v, float slope) { return v > 0.f ? v : v * slope; }

// 16 bytes of values through act(); bf16 pairs are widened, scaled and rounded to nearest even
template <typename T>
__device__ __forceinline__ u32x4 act16(u32x4 v, float slope) {
  if constexpr (sizeof(T) == 4) {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = __float_as_uint(act1(__uint_as_float(v[e]), slope));
    return r;
  } else {
    const f32x4 lo = widen_bf16x4(u32x2{v[0], v[1]}), hi = widen_bf16x4(u32x2{v[2], v[3]});
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = act1(lo[e], slope); b[e] = act1(hi[e], slope); }
    const u32x2 ra = cvt_bf16x4(a), rb = cvt_bf16x4(b);
    return u32x4{ra[0], ra[1], rb[0], rb[1]};
  }
}

// g * (y > 0 ? 1 : slope), 16 bytes at a time
template <typename T>
__device__ __forceinline__ u32x4 mask16(u32x4 g, u32x4 y, float slope) {
  if constexpr (sizeof(T) == 4) {
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = __float_as_uint(__uint_as_float(y[e]) > 0.f ? __uint_as_float(g[e]) : __uint_as_float(g[e]) * slope);
    return r;
  } else {
    const f32x4 glo = widen_bf16x4(u32x2{g[0], g[1]}), ghi = widen_bf16x4(u32x2{g[2], g[3]});
    const f32x4 ylo = widen_bf16x4(u32x2{y[0], y[1]}), yhi = widen_bf16x4(u32x2{y[2], y[3]});
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] = ylo[e] > 0.f ? glo[e] : glo[e] * slope; b[e] = yhi[e] > 0.f ? ghi[e] : ghi[e] * slope; }
    const u32x2 ra = cvt_bf16x4(a), rb = cvt_bf16x4(b);
    return u32x4{ra[0], ra[1], rb[0], rb[1]};
  }
}

// va / vb: 16-byte vectors per pixel of a / b; total = P * (va + vb) output vectors
template <typename T>
__global__ __launch_bounds__(kThreads) void act_concat_fwd_kernel(const u32x4* a, const u32x4* b, u32x4* y, uint32_t total, uint32_t va,
                                                                  uint32_t vb, float slope) {
  const uint32_t vy = va + vb;
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
    const uint32_t p = i / vy, v = i - p * vy;
    const u32x4 s = v < va ? a[(size_t)p * va + v] : b[(size_t)p * vb + (v - va)];
    y[i] = act16<T>(s, slope);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void act_concat_bwd_kernel(const u32x4* g, const u32x4* y, u32x4* ga, u32x4* gb, uint32_t total,
                                                                  uint32_t va, uint32_t vb, float slope) {
  const uint32_t vy = va + vb;
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total; i += gridDim.x * kThreads) {
    const uint32_t p = i / vy, v = i - p * vy;
    const u32x4 r = mask16<T>(g[i], y[i], slope);
    if (v < va) ga[(size_t)p * va + v] = r;
    else gb[(size_t)p * vb + (v - va)] = r;
  }
}

// common argument checks; on success *total = number of 16-byte output vectors, *va / *vb = vectors per pixel
bool concat_shape(const char* who, int64_t P, int32_t Ca, int32_t Cb, int32_t dtype, uint32_t* total, uint32_t* va, uint32_t* vb) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("%s: dtype must be GC_F32 or GC_BF16", who); return false; }
  const int per = dtype == GC_F32 ? 4 : 8;
  if (P < 0 || Ca <= 0 || Cb <= 0 || Ca % per != 0 || Cb % per != 0) {
    set_error("%s: P >= 0 and channel counts that are positive multiples of %d expected (Ca=%d, Cb=%d)", who, per, Ca, Cb);
    return false;
  }
  const int64_t vecs = P * (int64_t)((Ca + Cb) / per);
  if (vecs >= (1ll << 31)) { set_error("%s: tensor too large (more than 2^31 16-byte vectors)", who); return false; }
  *total = (uint32_t)vecs; *va = (uint32_t)(Ca / per); *vb = (uint32_t)(Cb / per);
  return true;
}

bool bad_ptr(const char* who, std::initializer_list<const void*> ptrs) {
  for (const void* q : ptrs)
    if (q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15u)) { set_error("%s: null or not 16-byte aligned pointer", who); return true; }
  return false;
}

unsigned concat_grid(uint32_t total) {
  unsigned blocks = (total + kThreads - 1) / kThreads;
  return blocks > 8192 ? 8192 : blocks;
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_act_concat_fwd(const void* a, const void* b, void* y, int64_t P, int32_t Ca, int32_t Cb, int32_t dtype, float slope,
                                 void* stream) {
  uint32_t total, va, vb;
  if (!concat_shape("gc_act_concat_fwd", P, Ca, Cb, dtype, &total, &va, &vb)) return GC_BAD_ARGUMENT;
  if (total == 0) return GC_OK;
  if (bad_ptr("gc_act_concat_fwd", {a, b, y})) return GC_BAD_ARGUMENT;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("gc::act_concat_fwd_kernel", 0.0, st);
  const u32x4 *pa = reinterpret_cast<const u32x4*>(a), *pb = reinterpret_cast<const u32x4*>(b);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(act_concat_fwd_kernel<float>, dim3(concat_grid(total)), dim3(kThreads), 0, st, pa, pb, reinterpret_cast<u32x4*>(y), total, va, vb, slope);
  else
    hipLaunchKernelGGL(act_concat_fwd_kernel<uint16_t>, dim3(concat_grid(total)), dim3(kThreads), 0, st, pa, pb, reinterpret_cast<u32x4*>(y), total, va, vb, slope);
  return check_hip(hipGetLastError(), "act_concat_fwd launch");
}

extern "C" int gc_act_concat_bwd(const void* g, const void* y, void* ga, void* gb, int64_t P, int32_t Ca, int32_t Cb, int32_t dtype,
                                 float slope, void* stream) {
  uint32_t total, va, vb;
  if (!concat_shape("gc_act_concat_bwd", P, Ca, Cb, dtype, &total, &va, &vb)) return GC_BAD_ARGUMENT;
  if (total == 0) return GC_OK;
  if (bad_ptr("gc_act_concat_bwd", {g, y, ga, gb})) return GC_BAD_ARGUMENT;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("gc::act_concat_bwd_kernel", 0.0, st);
  const u32x4 *pg = reinterpret_cast<const u32x4*>(g), *py = reinterpret_cast<const u32x4*>(y);
  u32x4 *qa = reinterpret_cast<u32x4*>(ga), *qb = reinterpret_cast<u32x4*>(gb);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(act_concat_bwd_kernel<float>, dim3(concat_grid(total)), dim3(kThreads), 0, st, pg, py, qa, qb, total, va, vb, slope);
  else
    hipLaunchKernelGGL(act_concat_bwd_kernel<uint16_t>, dim3(concat_grid(total)), dim3(kThreads), 0, st, pg, py, qa, qb, total, va, vb, slope);
  return check_hip(hipGetLastError(), "act_concat_bwd launch");
}
