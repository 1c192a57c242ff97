// Adam step over all parameters of an optimizer in two launches (a counter increment and one multi-tensor update),
// with the arithmetic of torch.optim.Adam (amsgrad = false, weight_decay = 0, maximize = false), fp32:
//   m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// Every parameter has its own step count t, as in torch.optim.Adam (a parameter that receives its first gradient late
// starts its bias correction at t = 1); the counts live in device memory so that the call can be replayed inside a HIP
// graph.
// Optionally the same launch keeps an exponential moving average of the parameters (the reference's averaged generator,
// StyleGAN/train.py:71-76, there a separate pass over both parameter sets after the step):  e = w e + (1 - w) p_new.
#include "common.h"

namespace gc {

constexpr int kAdamMaxTensors = 48;
constexpr int kAdamChunk = kThreads * 4 * 4;   // elements per workgroup: four 16-byte accesses per thread and tensor

struct AdamTable {
  float* p[kAdamMaxTensors];
  const float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  float* e[kAdamMaxTensors];                  // moving average of p, or null
  float* step[kAdamMaxTensors];               // this parameter's step count (one float)
  int64_t numel[kAdamMaxTensors];
  int32_t first_block[kAdamMaxTensors + 1];   // prefix sum of workgroups per tensor
  int32_t n;
};

__global__ void adam_tick_kernel(const AdamTable t) {
  if ((int)threadIdx.x < t.n) t.step[threadIdx.x][0] += 1.f;
}

__global__ __launch_bounds__(kThreads) void adam_kernel(const AdamTable t, float lr, float b1, float b2, float eps, float ema_w) {
  int ti = 0;
  while (ti + 1 < t.n && (int)blockIdx.x >= t.first_block[ti + 1]) ti++;
  const int64_t base = (int64_t)(blockIdx.x - t.first_block[ti]) * kAdamChunk;
  const int64_t n = t.numel[ti];
  float* __restrict__ p = t.p[ti];
  const float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  float* __restrict__ e = t.e[ti];
  const float s = t.step[ti][0];
  const float bc1 = 1.f - powf(b1, s), bc2 = 1.f - powf(b2, s);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  auto update = [&](float gi, float m0, float v0, float p0, float& mi, float& vi, float& pi) {
    const float w = 1.f - b1;                                      // lerp_(grad, 1 - beta1), torch's two-sided formula:
    mi = w < 0.5f ? m0 + w * (gi - m0) : gi - (gi - m0) * (1.f - w);   // exact at both ends (beta1 = 0 gives g)
    vi = v0 * b2 + (1.f - b2) * gi * gi;                           // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = p0 - step_size * (mi / denom);                            // addcdiv_(exp_avg, denom, value = -step_size)
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(e)) & 15u) == 0;
  if (vec && base + kAdamChunk <= n) {       // a full chunk of 16-byte aligned tensors: four elements per access
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int64_t i = base + ((int64_t)k * kThreads + threadIdx.x) * 4;
      const f32x4 gi = *reinterpret_cast<const f32x4*>(g + i), m0 = *reinterpret_cast<const f32x4*>(m + i);
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(v + i), p0 = *reinterpret_cast<const f32x4*>(p + i);
      f32x4 mi, vi, pi;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float a, b, c;
        update(gi[q], m0[q], v0[q], p0[q], a, b, c);
        mi[q] = a; vi[q] = b; pi[q] = c;
      }
      *reinterpret_cast<f32x4*>(m + i) = mi;
      *reinterpret_cast<f32x4*>(v + i) = vi;
      *reinterpret_cast<f32x4*>(p + i) = pi;
      if (e != nullptr) {
        f32x4 ei = *reinterpret_cast<const f32x4*>(e + i);
#pragma unroll
        for (int q = 0; q < 4; q++) ei[q] = ei[q] * ema_w + (1.f - ema_w) * pi[q];     // mul_(weight).add_(1 - weight, p)
        *reinterpret_cast<f32x4*>(e + i) = ei;
      }
    }
    return;
  }
  for (int k = 0; k < 16; k++) {
    const int64_t i = base + (int64_t)k * kThreads + threadIdx.x;
    if (i >= n) break;
    float mi, vi, pi;
    update(g[i], m[i], v[i], p[i], mi, vi, pi);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (e != nullptr) e[i] = e[i] * ema_w + (1.f - ema_w) * pi;
  }
}

}  // namespace gc

using namespace gc;

static int adam_launch(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       float* const* ema, const int64_t* numel, float* const* steps, float lr, float beta1, float beta2, float eps,
                       float ema_w, void* stream, const char* who) {
  if (n < 0 || (n > 0 && (params == nullptr || grads == nullptr || exp_avg == nullptr || exp_avg_sq == nullptr || numel == nullptr ||
                          steps == nullptr))) {
    set_error("%s: null pointer", who); return GC_BAD_ARGUMENT;
  }
  static_assert(kAdamMaxTensors <= 64, "the counter increment is one wavefront");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps("gc::adam_kernel", 0.0, s);
  for (int i0 = 0; i0 < n; i0 += kAdamMaxTensors) {
    AdamTable t;
    t.n = n - i0 < kAdamMaxTensors ? n - i0 : kAdamMaxTensors;
    int blocks = 0;
    for (int j = 0; j < t.n; j++) {
      const int i = i0 + j;
      if (numel[i] < 0 || params[i] == nullptr || grads[i] == nullptr || exp_avg[i] == nullptr || exp_avg_sq[i] == nullptr ||
          steps[i] == nullptr) {
        set_error("%s: bad tensor %d", who, i); return GC_BAD_ARGUMENT;
      }
      t.p[j] = params[i]; t.g[j] = grads[i]; t.m[j] = exp_avg[i]; t.v[j] = exp_avg_sq[i]; t.numel[j] = numel[i];
      t.e[j] = ema != nullptr ? ema[i] : nullptr;
      t.step[j] = steps[i];
      t.first_block[j] = blocks;
      blocks += (int)((numel[i] + kAdamChunk - 1) / kAdamChunk);
    }
    t.first_block[t.n] = blocks;
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, s, t);
    if (blocks > 0) hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, s, t, lr, beta1, beta2, eps, ema_w);
  }
  return check_hip(hipGetLastError(), "adam launch");
}

extern "C" int gc_adam_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                            const int64_t* numel, float* const* steps, float lr, float beta1, float beta2, float eps, void* stream) {
  return adam_launch(n, params, grads, exp_avg, exp_avg_sq, nullptr, numel, steps, lr, beta1, beta2, eps, 0.f, stream, "gc_adam_step");
}

extern "C" int gc_adam_ema_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                                float* const* exp_avg_sq, float* const* ema, float ema_weight, const int64_t* numel,
                                float* const* steps, float lr, float beta1, float beta2, float eps, void* stream) {
  if (ema == nullptr || !(ema_weight >= 0.f && ema_weight <= 1.f)) {
    set_error("gc_adam_ema_step: null table of averages or weight outside [0, 1]"); return GC_BAD_ARGUMENT;
  }
  return adam_launch(n, params, grads, exp_avg, exp_avg_sq, ema, numel, steps, lr, beta1, beta2, eps, ema_weight, stream, "gc_adam_ema_step");
}
