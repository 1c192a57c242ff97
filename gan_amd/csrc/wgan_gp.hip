// The scalar parts of a WGAN-GP critic iteration (BasicGANs/loss_functions.py:7-27) as a handful of launches instead of
// the ~50 elementwise and reduction launches the same expressions cost as framework tensor operations:
//   gc_wgan_mix                 mixed = real * eps + fake * (1 - eps)                                  (:19)
//   gc_gradient_penalty_fwd     lambda * mean((||gradient_n||_2 - 1)^2), and the factor each sample's gradient is
//                               multiplied by in the backward pass                                    (:26-27)
//   gc_gradient_penalty_bwd     g_penalty * factor_n * gradient_n
//   gc_wgan_critic_loss         mean(D(x)) - drift * mean(D(x)^2), mean(D(G(z))) and the gradient of
//                               (fake term - real term) with respect to the 2B scores                 (:7-15)
// All of it is HBM- or latency-bound: 16-byte accesses, fixed summation order (results do not depend on timing).
#include "common.h"

namespace gc {
namespace {

constexpr int kMaxChunks = 8;

__device__ __forceinline__ float block_sum(float v, float* red) {   // red: kThreads / 64 floats of LDS; all threads get the sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();            // (red may still be read from a previous call)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) s += red[w];
  return s;
}

// one thread: 4 consecutive elements; sample = element / per_sample
__global__ __launch_bounds__(kThreads) void wgan_mix_kernel(const float* real, const float* fake, const float* eps, float* out,
                                                             uint32_t total4, uint32_t per_sample4) {
#pragma clang fp contract(off)      // two products and a sum, as the reference's expression rounds them
  for (uint32_t e = blockIdx.x * kThreads + threadIdx.x; e < total4; e += gridDim.x * kThreads) {
    const float a = eps[e / per_sample4];
    const float b = 1.f - a;
    const f32x4 r = reinterpret_cast<const f32x4*>(real)[e], f = reinterpret_cast<const f32x4*>(fake)[e];
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; i++) o[i] = r[i] * a + f[i] * b;
    reinterpret_cast<f32x4*>(out)[e] = o;
  }
}

// partial[n * chunks + c] = sum of squares of chunk c of sample n
template <typename T>
__global__ __launch_bounds__(kThreads) void gp_sumsq_kernel(const void* g, float* partial, uint32_t per_sample4, uint32_t chunk4) {
  __shared__ float red[kThreads / 64];
  const uint32_t n = blockIdx.y, c = blockIdx.x;
  const uint32_t lo = c * chunk4, hi = min(per_sample4, lo + chunk4);
  const size_t base = (size_t)n * per_sample4;
  float s = 0.f;
  for (uint32_t e = lo + threadIdx.x; e < hi; e += kThreads) {
    const f32x4 v = load4<T>(g, (base + e) * 4);
    s += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[n * gridDim.x + c] = s;
}

// one block: norms, backward factors and the penalty
__global__ __launch_bounds__(kThreads) void gp_finish_kernel(const float* partial, float* norms, float* coef, float* penalty, int N,
                                                              int chunks, float lambda) {
  __shared__ float red[kThreads / 64];
  float acc = 0.f;
  for (int n = threadIdx.x; n < N; n += kThreads) {
    float s = 0.f;
    for (int c = 0; c < chunks; c++) s += partial[n * chunks + c];
    const float nm = sqrtf(s);
    const float d = nm - 1.f;
    norms[n] = nm;
    // d/dg of lambda / N * (||g|| - 1)^2 = lambda * 2 (||g|| - 1) / (N ||g||) * g; zero at g = 0 (the subgradient torch's norm takes)
    coef[n] = nm > 0.f ? lambda * 2.f * d / ((float)N * nm) : 0.f;
    acc += d * d;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) penalty[0] = lambda * (acc / (float)N);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gp_bwd_kernel(const void* g, const float* coef, const float* gout, void* out, uint32_t total4,
                                                           uint32_t per_sample4) {
  const float go = gout[0];
  for (uint32_t e = blockIdx.x * kThreads + threadIdx.x; e < total4; e += gridDim.x * kThreads) {
    const float k = go * coef[e / per_sample4];
    store4<T>(out, (size_t)e * 4, load4<T>(g, (size_t)e * 4) * k);
  }
}

// one block.  scores: B real-batch scores followed by B generated-batch scores.
__global__ __launch_bounds__(kThreads) void wgan_critic_loss_kernel(const float* scores, float* out, float* grad, int B, float drift) {
  __shared__ float red[kThreads / 64];
  float sr = 0.f, sq = 0.f, sf = 0.f;
  const float inv = 1.f / (float)B;
  for (int i = threadIdx.x; i < B; i += kThreads) {
    const float r = scores[i], f = scores[B + i];
    sr += r;
    sq += r * r;
    sf += f;
    grad[i] = (2.f * drift * r - 1.f) * inv;     // d(fake term - real term) / d score
    grad[B + i] = inv;
  }
  sr = block_sum(sr, red);
  sq = block_sum(sq, red);
  sf = block_sum(sf, red);
  if (threadIdx.x == 0) {
    out[0] = sr * inv - drift * (sq * inv);
    out[1] = sf * inv;
  }
}

unsigned grid_for4(uint64_t total4) {
  const uint64_t blocks = (total4 + kThreads - 1) / kThreads;
  return (unsigned)(blocks < 256 * 16 ? (blocks ? blocks : 1) : 256 * 16);
}

bool aligned16(const void* p) { return p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_batch(const char* what, int32_t N, int64_t per_sample) {
  if (N <= 0 || per_sample <= 0 || per_sample % 4 != 0 || (uint64_t)N * (uint64_t)per_sample >= (1ull << 33)) {
    set_error("%s: need N > 0, elements per sample a positive multiple of 4 and fewer than 2^33 elements (got N=%d, %lld per sample)", what,
              N, (long long)per_sample);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  return GC_OK;
}

int chunks_for(int64_t per_sample) {
  const int64_t c = (per_sample / 4 + 1023) / 1024;      // at least 1024 16-byte loads per block
  return (int)(c < 1 ? 1 : (c > kMaxChunks ? kMaxChunks : c));
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" {

int gc_wgan_mix(const float* real, const float* fake, const float* eps, float* out, int32_t N, int64_t per_sample, void* stream) {
  if (int rc = check_batch("gc_wgan_mix", N, per_sample)) return rc;
  if (!aligned16(real) || !aligned16(fake) || !aligned16(out) || eps == nullptr) {
    set_error("gc_wgan_mix: tensors must be non-null and 16-byte aligned");
    return GC_BAD_ARGUMENT;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint64_t total4 = (uint64_t)N * (uint64_t)per_sample / 4;
  ProfScope ps("gc::wgan_mix_kernel", 0.0, s);
  hipLaunchKernelGGL(wgan_mix_kernel, dim3(grid_for4(total4)), dim3(kThreads), 0, s, real, fake, eps, out, (uint32_t)total4,
                     (uint32_t)(per_sample / 4));
  return check_hip(hipGetLastError(), "wgan_mix launch");
}

size_t gc_gradient_penalty_workspace_bytes(int32_t N, int64_t per_sample) {
  if (N <= 0 || per_sample <= 0) return 0;
  return (size_t)N * (size_t)chunks_for(per_sample) * sizeof(float);
}

int gc_gradient_penalty_fwd(const void* gradient, int32_t dtype, int32_t N, int64_t per_sample, float lambda, float* norms, float* coef,
                            float* penalty, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_batch("gc_gradient_penalty_fwd", N, per_sample)) return rc;
  if (!aligned16(gradient) || norms == nullptr || coef == nullptr || penalty == nullptr || (dtype != GC_F32 && dtype != GC_BF16)) {
    set_error("gc_gradient_penalty_fwd: gradient must be 16-byte aligned fp32 or bf16, the outputs non-null");
    return GC_BAD_ARGUMENT;
  }
  if (N > 65535) {
    set_error("gc_gradient_penalty_fwd: at most 65535 samples (got %d)", N);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  const size_t need = gc_gradient_penalty_workspace_bytes(N, per_sample);
  if (workspace == nullptr || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 3) != 0) {
    set_error("gc_gradient_penalty_fwd: workspace of %zu bytes needed (got %zu)", need, workspace_bytes);
    return GC_WORKSPACE_TOO_SMALL;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int chunks = chunks_for(per_sample);
  const uint32_t ps4 = (uint32_t)(per_sample / 4), chunk4 = (ps4 + chunks - 1) / chunks;
  float* partial = reinterpret_cast<float*>(workspace);
  {
    ProfScope ps("gc::gp_sumsq_kernel", 0.0, s);
    if (dtype == GC_F32) hipLaunchKernelGGL(gp_sumsq_kernel<float>, dim3(chunks, N), dim3(kThreads), 0, s, gradient, partial, ps4, chunk4);
    else hipLaunchKernelGGL(gp_sumsq_kernel<uint16_t>, dim3(chunks, N), dim3(kThreads), 0, s, gradient, partial, ps4, chunk4);
  }
  if (int rc = check_hip(hipGetLastError(), "gp_sumsq launch")) return rc;
  ProfScope ps("gc::gp_finish_kernel", 0.0, s);
  hipLaunchKernelGGL(gp_finish_kernel, dim3(1), dim3(kThreads), 0, s, partial, norms, coef, penalty, N, chunks, lambda);
  return check_hip(hipGetLastError(), "gp_finish launch");
}

int gc_gradient_penalty_bwd(const void* gradient, int32_t dtype, int32_t N, int64_t per_sample, const float* coef, const float* g_penalty,
                            void* out, void* stream) {
  if (int rc = check_batch("gc_gradient_penalty_bwd", N, per_sample)) return rc;
  if (!aligned16(gradient) || !aligned16(out) || coef == nullptr || g_penalty == nullptr || (dtype != GC_F32 && dtype != GC_BF16)) {
    set_error("gc_gradient_penalty_bwd: tensors must be non-null, 16-byte aligned, fp32 or bf16");
    return GC_BAD_ARGUMENT;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const uint64_t total4 = (uint64_t)N * (uint64_t)per_sample / 4;
  ProfScope ps("gc::gp_bwd_kernel", 0.0, s);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(gp_bwd_kernel<float>, dim3(grid_for4(total4)), dim3(kThreads), 0, s, gradient, coef, g_penalty, out, (uint32_t)total4,
                       (uint32_t)(per_sample / 4));
  else
    hipLaunchKernelGGL(gp_bwd_kernel<uint16_t>, dim3(grid_for4(total4)), dim3(kThreads), 0, s, gradient, coef, g_penalty, out,
                       (uint32_t)total4, (uint32_t)(per_sample / 4));
  return check_hip(hipGetLastError(), "gp_bwd launch");
}

int gc_wgan_critic_loss(const float* scores, int32_t B, float drift, float* out, float* grad, void* stream) {
  if (scores == nullptr || out == nullptr || grad == nullptr || B <= 0) {
    set_error("gc_wgan_critic_loss: scores [2B], out [2] and grad [2B] must be non-null, B > 0");
    return GC_BAD_ARGUMENT;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  ProfScope ps("gc::wgan_critic_loss_kernel", 0.0, s);
  hipLaunchKernelGGL(wgan_critic_loss_kernel, dim3(1), dim3(kThreads), 0, s, scores, out, grad, B, drift);
  return check_hip(hipGetLastError(), "wgan_critic_loss launch");
}

}  // extern "C"
