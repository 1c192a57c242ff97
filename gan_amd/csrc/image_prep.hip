// Input side of the training step: decoded 8-bit images (rows of interleaved channels, as cv2.imread returns them) to the
// float batch the networks take, in one launch per batch.  Replaces, per sample, the host-side chain of the reference's
// datasets (BasicGANs/utils.py:46-55, Pix2Pix/dataset.py:225-318, StyleGAN/train.py:30-38):
//     cv2.resize(image, size [, INTER_LINEAR | INTER_NEAREST])  ->  jitter crop  ->  cv2.flip(image, 1)
//     ->  float / 255  ->  Normalize(mean, std)  ->  channel-first tensor
// The bilinear resize follows OpenCV's 8-bit path: source coordinate (d + 0.5) * scale - 0.5 with the two border clamps,
// weights rounded to 11-bit fixed point, horizontal pass in integers, vertical pass
// (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, result an 8-bit value; nearest takes
// min(floor(d * scale), size - 1).  The float steps are the reference's, in its order and in fp32: value / 255, then
// (value - mean) / std.  HBM-bound byte work: one thread per output pixel, all channels; reads hit the caches (a 2 x 2
// neighbourhood per output pixel), writes are contiguous along the output row in every plane.
#include "common.h"

namespace gc {
namespace {

constexpr int kCoefBits = 11, kCoefScale = 1 << kCoefBits;

struct AxisTap {
  int s0, s1;        // the two source indices
  int a0, a1;        // their 11-bit weights
};

// OpenCV's coordinate and weight of destination index d (resize.cpp: the INTER_LINEAR branch of cv::resize)
__device__ __forceinline__ AxisTap linear_tap(int d, int dst_size, int src_size) {
  const double scale = (double)src_size / dst_size;
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src_size - 1) { f = 0.f; s = src_size - 1; }
  AxisTap t;
  t.s0 = s;
  t.s1 = min(s + 1, src_size - 1);
  // saturate_cast<short>(w * 2048): round to nearest, ties to even
  t.a0 = (int)rintf((1.f - f) * kCoefScale);
  t.a1 = (int)rintf(f * kCoefScale);
  return t;
}

__device__ __forceinline__ int nearest_index(int d, int dst_size, int src_size) {
  const double inv = (double)src_size / dst_size;
  return min((int)floor(d * inv), src_size - 1);
}

template <typename TOut>
__global__ __launch_bounds__(kThreads) void image_prepare_kernel(const gc_image_job* jobs, int channels, int out_h, int out_w, int interp,
                                                                 float divisor, f32x4 mean, f32x4 stdv, TOut* out, long sn, long sc,
                                                                 long sh, long sw) {
  const int n = blockIdx.y;
  const gc_image_job job = jobs[n];
  const uint8_t* src = static_cast<const uint8_t*>(job.src);
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < out_h * out_w; p += gridDim.x * kThreads) {
    const int oy = p / out_w, ox = p - oy * out_w;
    // output pixel -> pixel of the resized image: crop window, then the horizontal flip of the cropped image
    // (clamped into the resized image: whatever a job says, every source index below stays inside its image)
    const int ry = min(max(job.crop_y + oy, 0), job.resize_h - 1);
    const int rx = min(max(job.crop_x + (job.flip ? out_w - 1 - ox : ox), 0), job.resize_w - 1);
    int v[4] = {0, 0, 0, 0};
    if (interp == 0) {
      const int sy = nearest_index(ry, job.resize_h, job.src_h), sx = nearest_index(rx, job.resize_w, job.src_w);
      const uint8_t* q = src + (size_t)sy * job.src_stride + (size_t)sx * channels;
      for (int c = 0; c < channels; c++) v[c] = q[c];
    } else {
      const AxisTap ty = linear_tap(ry, job.resize_h, job.src_h), tx = linear_tap(rx, job.resize_w, job.src_w);
      const uint8_t* r0 = src + (size_t)ty.s0 * job.src_stride;
      const uint8_t* r1 = src + (size_t)ty.s1 * job.src_stride;
      for (int c = 0; c < channels; c++) {
        const int h0 = r0[tx.s0 * channels + c] * tx.a0 + r0[tx.s1 * channels + c] * tx.a1;
        const int h1 = r1[tx.s0 * channels + c] * tx.a0 + r1[tx.s1 * channels + c] * tx.a1;
        v[c] = (((ty.a0 * (h0 >> 4)) >> 16) + ((ty.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
      }
    }
    TOut* dst = out + (size_t)n * sn + (size_t)oy * sh + (size_t)ox * sw;
    for (int c = 0; c < channels; c++) {
      float x = (float)v[c];
      if (divisor != 1.f) x = x / divisor;               // the reference divides (by 255), it does not multiply by a reciprocal
      x = (x - mean[c]) / stdv[c];
      if (sizeof(TOut) == 4) reinterpret_cast<float*>(dst)[(size_t)c * sc] = x;
      else reinterpret_cast<uint16_t*>(dst)[(size_t)c * sc] = float_to_bf16_bits(x);
    }
  }
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_image_batch_prepare(const gc_image_job* jobs, int32_t n, int32_t channels, int32_t out_h, int32_t out_w, int32_t interp,
                                      float divisor, const float* mean, const float* std_, void* out, int32_t out_dtype,
                                      const int64_t out_strides[4], void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (n == 0) return GC_OK;
  if (jobs == nullptr || out == nullptr || out_strides == nullptr || n < 0 || n > 65535 || channels < 1 || channels > 4 || out_h < 1 || out_w < 1 ||
      (long)out_h * out_w >= (1l << 31) || (interp != 0 && interp != 1) || !(divisor > 0.f) ||
      (out_dtype != GC_F32 && out_dtype != GC_BF16)) {
    set_error("gc_image_batch_prepare: null pointer, more than 65535 images or 4 channels, empty output, unknown interpolation / dtype or divisor <= 0");
    return GC_BAD_ARGUMENT;
  }
  f32x4 m = {0.f, 0.f, 0.f, 0.f}, s = {1.f, 1.f, 1.f, 1.f};
  for (int c = 0; c < channels; c++) {
    if (mean != nullptr) m[c] = mean[c];
    if (std_ != nullptr) s[c] = std_[c];
    if (!(s[c] != 0.f)) {
      set_error("gc_image_batch_prepare: std[%d] is zero", c);
      return GC_BAD_ARGUMENT;
    }
  }
  const int per_image = (int)(((long)out_h * out_w + kThreads - 1) / kThreads);
  const dim3 grid((unsigned)(per_image > 64 ? 64 : per_image), (unsigned)n);
  ProfScope ps("gc::image_prepare_kernel", 0.0, stream);
  if (out_dtype == GC_F32)
    hipLaunchKernelGGL(image_prepare_kernel<float>, grid, dim3(kThreads), 0, stream, jobs, channels, out_h, out_w, interp, divisor, m, s,
                       static_cast<float*>(out), (long)out_strides[0], (long)out_strides[1], (long)out_strides[2], (long)out_strides[3]);
  else
    hipLaunchKernelGGL(image_prepare_kernel<uint16_t>, grid, dim3(kThreads), 0, stream, jobs, channels, out_h, out_w, interp, divisor, m, s,
                       static_cast<uint16_t*>(out), (long)out_strides[0], (long)out_strides[1], (long)out_strides[2], (long)out_strides[3]);
  return check_hip(hipGetLastError(), "image_prepare launch");
}
