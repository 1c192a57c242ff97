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
constexpr int kAdamChunk = kThreads * 4 * 4;   // elements per workgroup

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
  for (int k = 0; k < 16; k++) {
    const int64_t i = base + (int64_t)k * kThreads + threadIdx.x;
    if (i >= n) break;
    const float gi = g[i];
    const float w = 1.f - b1, m0 = m[i];                          // lerp_(grad, 1 - beta1), torch's two-sided formula:
    const float mi = w < 0.5f ? m0 + w * (gi - m0) : gi - (gi - m0) * (1.f - w);   // exact at both ends (beta1 = 0 gives g)
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;           // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    const float pi = p[i] - step_size * (mi / denom);            // addcdiv_(exp_avg, denom, value = -step_size)
    p[i] = pi;
    if (e != nullptr) e[i] = e[i] * ema_w + (1.f - ema_w) * pi;  // mul_(weight).add_(1 - weight, p)
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
