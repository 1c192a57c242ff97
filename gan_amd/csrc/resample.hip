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
// `addend` (null, or a tensor of y's shape): y = scale * pool(x) + addend, the residual sum of a block in the same sweep
template <typename T>
__global__ __launch_bounds__(kThreads) void pool2x2_kernel(const T* x, T* y, size_t total, int H, int W, int C8, float scale, const T* addend) {
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
    if (addend != nullptr) {
      const Vec8<T> r = Vec8<T>::load(addend + e * 8);
      o.a += r.a;
      o.b += r.b;
    }
    o.store(y + e * 8);
  }
}

// y[n, 2h + a, 2w + b, :] = scale * x[n, h, w, :];  H, W: extents of x.  One thread per input element, four stores.
template <typename T>
__global__ __launch_bounds__(kThreads) void replicate2x_kernel(const T* x, T* y, size_t total, int H, int W, int C8, float scale, const T* addend) {
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
    if (addend != nullptr) {          // y = scale * replicate(x) + addend (addend has y's shape)
      const size_t offs[4] = {dst, dst + C8, dst + rowp, dst + rowp + C8};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        Vec8<T> r = Vec8<T>::load(addend + offs[q] * 8);
        r.a += v.a;
        r.b += v.b;
        r.store(y + offs[q] * 8);
      }
      continue;
    }
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

// Reflection padding by `pad` pixels on all four sides (torch.nn.ReflectionPad2d; reference Pix2Pix/models.py:162,185,
// 204,210: in front of the 7x7 and 3x3 convolutions of the Pix2PixHD generator), NHWC, and its adjoint:
//   forward   y[n, oh, ow, :] = x[n, r(oh - pad, H), r(ow - pad, W), :],  r(i, L) = i < 0 ? -i : (i >= L ? 2 (L - 1) - i : i)
//   adjoint   gx[n, h, w, :]  = sum of gy over the (at most 3 x 3) padded positions that read (h, w)
// One thread moves 8 channels of one output pixel; H, W are the extents of the unpadded tensor in both directions.
__device__ __forceinline__ int reflect_index(int i, int L) { return i < 0 ? -i : (i >= L ? 2 * (L - 1) - i : i); }

// padded positions that read unpadded index h (extent L): h + pad always, pad - h for 1 <= h <= pad, and
// pad + 2 (L - 1) - h for L - 1 - pad <= h <= L - 2
__device__ __forceinline__ int reflect_sources(int h, int L, int pad, int (&src)[3]) {
  int n = 0;
  src[n++] = h + pad;
  if (h >= 1 && h <= pad) src[n++] = pad - h;
  if (h >= L - 1 - pad && h <= L - 2) src[n++] = pad + 2 * (L - 1) - h;
  return n;
}

template <bool ADJOINT, typename T>
__global__ __launch_bounds__(kThreads) void reflect_pad_kernel(const T* x, T* y, uint32_t total, int H, int W, int C8, int pad) {
  const uint32_t e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int HP = H + 2 * pad, WP = W + 2 * pad;
  const int OH = ADJOINT ? H : HP, OW = ADJOINT ? W : WP;          // extents of the tensor this launch writes
  const int c = (int)(e % (uint32_t)C8);
  uint32_t r = e / (uint32_t)C8;
  const int ow = (int)(r % (uint32_t)OW);
  r /= (uint32_t)OW;
  const int oh = (int)(r % (uint32_t)OH);
  const size_t n = r / (uint32_t)OH;
  if (!ADJOINT) {
    const int h = reflect_index(oh - pad, H), w = reflect_index(ow - pad, W);
    Vec8<T>::load(x + (((n * H + h) * W + w) * C8 + c) * 8).store(y + (size_t)e * 8);
  } else {
    int hs[3], ws[3];
    const int nh = reflect_sources(oh, H, pad, hs), nw = reflect_sources(ow, W, pad, ws);
    Vec8<T> acc;
    acc.a = f32x4{0.f, 0.f, 0.f, 0.f};
    acc.b = acc.a;
    for (int i = 0; i < nh; i++)
      for (int j = 0; j < nw; j++) {
        const Vec8<T> v = Vec8<T>::load(x + (((n * HP + hs[i]) * WP + ws[j]) * C8 + c) * 8);
        acc.a += v.a;
        acc.b += v.b;
      }
    acc.store(y + (size_t)e * 8);
  }
}

static int reflect_launch(bool adjoint, const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t dtype,
                          void* stream) {
  if (int e = check_args(x, y, N, H, W, C, dtype)) return e;
  if (pad < 0 || pad >= H || pad >= W) { set_error("reflect_pad: the padding (%d) must be smaller than the extents (%dx%d)", pad, H, W); return GC_BAD_ARGUMENT; }
  const size_t total = (size_t)N * (adjoint ? H : H + 2 * pad) * (adjoint ? W : W + 2 * pad) * (C / 8);
  if (total == 0) return GC_OK;
  if ((size_t)N * (H + 2 * pad) * (W + 2 * pad) * (C / 8) >= (1ull << 32) - kThreads) { set_error("reflect_pad: tensor too large"); return GC_UNSUPPORTED_GEOMETRY; }
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + kThreads - 1) / kThreads)), block(kThreads);
  ProfScope ps(adjoint ? "gc::reflect_pad_kernel<adjoint>" : "gc::reflect_pad_kernel", 0.0, s);
  if (dtype == GC_F32) {
    if (adjoint) hipLaunchKernelGGL((reflect_pad_kernel<true, float>), grid, block, 0, s, (const float*)x, (float*)y, (uint32_t)total, H, W, C / 8, pad);
    else hipLaunchKernelGGL((reflect_pad_kernel<false, float>), grid, block, 0, s, (const float*)x, (float*)y, (uint32_t)total, H, W, C / 8, pad);
  } else {
    if (adjoint) hipLaunchKernelGGL((reflect_pad_kernel<true, uint16_t>), grid, block, 0, s, (const uint16_t*)x, (uint16_t*)y, (uint32_t)total, H, W, C / 8, pad);
    else hipLaunchKernelGGL((reflect_pad_kernel<false, uint16_t>), grid, block, 0, s, (const uint16_t*)x, (uint16_t*)y, (uint32_t)total, H, W, C / 8, pad);
  }
  return check_hip(hipGetLastError(), "reflect_pad launch");
}

extern "C" int gc_reflect_pad(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t dtype, void* stream) {
  return reflect_launch(false, x, y, N, H, W, C, pad, dtype, stream);
}

extern "C" int gc_reflect_pad_adjoint(const void* gy, void* gx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t dtype,
                                      void* stream) {
  return reflect_launch(true, gy, gx, N, H, W, C, pad, dtype, stream);
}

static int pool_launch(const void* x, const void* addend, void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, float scale,
                       void* stream) {
  if (int e = check_args(x, y, N, Ho, Wo, C, dtype)) return e;
  if ((reinterpret_cast<uintptr_t>(addend) & 15) != 0) { set_error("resample: the addend must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  const size_t total = (size_t)N * Ho * Wo * (C / 8);
  if (total == 0) return GC_OK;
  ProfScope ps(addend != nullptr ? "gc::pool2x2_kernel<add>" : "gc::pool2x2_kernel", 0.0, (hipStream_t)stream);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(pool2x2_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float*)x, (float*)y, total, Ho, Wo, C / 8, scale, (const float*)addend);
  else
    hipLaunchKernelGGL(pool2x2_kernel<uint16_t>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, total, Ho, Wo, C / 8, scale, (const uint16_t*)addend);
  return check_hip(hipGetLastError(), "pool2x2 launch");
}

static int replicate_launch(const void* x, const void* addend, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, float scale,
                            void* stream) {
  if (int e = check_args(x, y, N, H, W, C, dtype)) return e;
  if ((reinterpret_cast<uintptr_t>(addend) & 15) != 0) { set_error("resample: the addend must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  const size_t total = (size_t)N * H * W * (C / 8);
  if (total == 0) return GC_OK;
  ProfScope ps(addend != nullptr ? "gc::replicate2x_kernel<add>" : "gc::replicate2x_kernel", 0.0, (hipStream_t)stream);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(replicate2x_kernel<float>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const float*)x, (float*)y, total, H, W, C / 8, scale, (const float*)addend);
  else
    hipLaunchKernelGGL(replicate2x_kernel<uint16_t>, dim3(grid_for(total)), dim3(kThreads), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, total, H, W, C / 8, scale, (const uint16_t*)addend);
  return check_hip(hipGetLastError(), "replicate2x launch");
}

extern "C" int gc_pool2x2(const void* x, void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, float scale, void* stream) {
  return pool_launch(x, nullptr, y, N, Ho, Wo, C, dtype, scale, stream);
}

extern "C" int gc_replicate2x(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, float scale, void* stream) {
  return replicate_launch(x, nullptr, y, N, H, W, C, dtype, scale, stream);
}

extern "C" int gc_pool2x2_add(const void* x, const void* addend, void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, float scale,
                              void* stream) {
  if (addend == nullptr) { set_error("resample: null addend"); return GC_BAD_ARGUMENT; }
  return pool_launch(x, addend, y, N, Ho, Wo, C, dtype, scale, stream);
}

extern "C" int gc_replicate2x_add(const void* x, const void* addend, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                                  float scale, void* stream) {
  if (addend == nullptr) { set_error("resample: null addend"); return GC_BAD_ARGUMENT; }
  return replicate_launch(x, addend, y, N, H, W, C, dtype, scale, stream);
}
