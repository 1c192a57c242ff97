// 2x2 sum-pooling and 2x nearest replication on NHWC tensors, each with a scale: the two are adjoint, so the pair
// covers AvgPool2d(2) (pool, 1/4) and its backward (replicate, 1/4), nearest Upsample(scale_factor=2) (replicate, 1)
// and its backward (pool, 1), to any order of differentiation.  Reference: BigGAN/model.py:117,188
// (Upsample / AvgPool2d in the residual blocks), StyleGAN/model.py:247,440 (nearest upsample, AvgPool2d).
// HBM-bound: one thread moves 8 channels (16 bytes of bf16, 32 of fp32) of one output pixel.
#include "common.h"

namespace gc {
namespace {

template <typename T>
struct Vec8 {
  f32x4 a, b;
  __device__ __forceinline__ static Vec8 load(const T* p) {
    Vec8 v;
    if (sizeof(T) == 4) {
      v.a = *reinterpret_cast<const f32x4*>(p);
      v.b = *reinterpret_cast<const f32x4*>(p + 4);
    } else {
      const u32x4 r = *reinterpret_cast<const u32x4*>(p);
      v.a = widen_bf16x4(u32x2{r[0], r[1]});
      v.b = widen_bf16x4(u32x2{r[2], r[3]});
    }
    return v;
  }
  __device__ __forceinline__ void store(T* p) const {
    if (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(p) = a;
      *reinterpret_cast<f32x4*>(p + 4) = b;
    } else {
      const u32x2 lo = cvt_bf16x4(a), hi = cvt_bf16x4(b);
      *reinterpret_cast<u32x4*>(p) = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
};

// y[n, h, w, :] = scale * sum of x[n, 2h + {0,1}, 2w + {0,1}, :];  H, W: extents of y
template <typename T>
__global__ __launch_bounds__(kThreads) void pool2x2_kernel(const T* x, T* y, size_t total, int H, int W, int C8, float scale) {
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (size_t)gridDim.x * kThreads) {
    const int c = (int)(e % C8);
    const size_t pix = e / C8;
    const int w = (int)(pix % W);
    const size_t nh = pix / W;                        // n * H + h
    const size_t src = ((nh * 2) * (2 * (size_t)W) + 2 * w) * C8 + c;   // row 2 (n H + h) of the [N * 2H, 2W] input
    const size_t rowp = 2 * (size_t)W * C8;
    const Vec8<T> v0 = Vec8<T>::load(x + src * 8), v1 = Vec8<T>::load(x + (src + C8) * 8);
    const Vec8<T> v2 = Vec8<T>::load(x + (src + rowp) * 8), v3 = Vec8<T>::load(x + (src + rowp + C8) * 8);
    Vec8<T> o;
    o.a = ((v0.a + v1.a) + (v2.a + v3.a)) * scale;
    o.b = ((v0.b + v1.b) + (v2.b + v3.b)) * scale;
    o.store(y + e * 8);
  }
}

// y[n, 2h + a, 2w + b, :] = scale * x[n, h, w, :];  H, W: extents of x.  One thread per input element, four stores.
template <typename T>
__global__ __launch_bounds__(kThreads) void replicate2x_kernel(const T* x, T* y, size_t total, int H, int W, int C8, float scale) {
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (size_t)gridDim.x * kThreads) {
    const int c = (int)(e % C8);
    const size_t pix = e / C8;
    const int w = (int)(pix % W);
    const size_t nh = pix / W;
    Vec8<T> v = Vec8<T>::load(x + e * 8);
    v.a *= scale;
    v.b *= scale;
    const size_t dst = ((nh * 2) * (2 * (size_t)W) + 2 * w) * C8 + c;
    const size_t rowp = 2 * (size_t)W * C8;
    v.store(y + dst * 8);
    v.store(y + (dst + C8) * 8);
    v.store(y + (dst + rowp) * 8);
    v.store(y + (dst + rowp + C8) * 8);
  }
}

int check_args(const void* x, const void* y, int64_t N, int32_t H, int32_t W, int32_t C, int32_t dtype) {
  if (x == nullptr || y == nullptr || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) != 0) {
    set_error("resample: tensors must be non-null and 16-byte aligned");
    return GC_BAD_ARGUMENT;
  }
  if (N < 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 != 0 || (dtype != GC_F32 && dtype != GC_BF16)) {
    set_error("resample: need C %% 8 == 0 and a gc_dtype (got N=%lld H=%d W=%d C=%d dtype=%d)", (long long)N, H, W, C, dtype);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  return GC_OK;
}

unsigned grid_for(size_t total) {
  const size_t blocks = (total + kThreads - 1) / kThreads;
  return (unsigned)(blocks < 256 * 64 ? (blocks ? blocks : 1) : 256 * 64);
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_pool2x2(const void* x, void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, float scale, void* stream) {
  if (int e = check_args(x, y, N, Ho, Wo, C, dtype)) return e;
  const size_t total = (size_t)N * Ho * Wo * (C / 8);
  if (total == 0) return GC_OK;
  ProfScope ps("gc::pool2x2_kernel", 0.0, (hipStream_t)stream);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(pool2x2_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float*)x, (float*)y, total, Ho, Wo, C / 8, scale);
  else
    hipLaunchKernelGGL(pool2x2_kernel<uint16_t>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, total, Ho, Wo, C / 8, scale);
  return check_hip(hipGetLastError(), "pool2x2 launch");
}

extern "C" int gc_replicate2x(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, float scale, void* stream) {
  if (int e = check_args(x, y, N, H, W, C, dtype)) return e;
  const size_t total = (size_t)N * H * W * (C / 8);
  if (total == 0) return GC_OK;
  ProfScope ps("gc::replicate2x_kernel", 0.0, (hipStream_t)stream);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(replicate2x_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float*)x, (float*)y, total, H, W, C / 8, scale);
  else
    hipLaunchKernelGGL(replicate2x_kernel<uint16_t>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, total, H, W, C / 8, scale);
  return check_hip(hipGetLastError(), "replicate2x launch");
}
