// BatchNorm2d (training-mode batch statistics or running statistics) fused with ReLU / LeakyReLU, NHWC fp32 or bf16
// activations (fp32 arithmetic, statistics and parameters):
// forward and first-order backward.
//
//   forward    y  = act(a * x + b),  a = gamma * r,  b = beta - mu * a,  mu / r = per-channel mean and
//              rsqrt(var + eps) over all N*H*W pixels (biased variance); running_mean / running_var (unbiased) and
//              num_batches_tracked are updated the way torch.nn.BatchNorm2d does
//   backward   g = gy * act'(a * x + b);  gbeta = sum g;  ggamma = sum g * xh;  gx = a * (g - gbeta/P - xh * ggamma/P)
//
// Three launches per direction, all bitwise reproducible (no float atomics):
//   1. partial statistics: a workgroup owns 64 channels x one slab of pixels and writes (mean, M2) of its slab
//      (forward: mean first, then centred squares, the second sweep of the slab comes from L2) or (sum g, sum g*xh);
//   2. finalize: one workgroup per 64 channels merges the slabs in a fixed order (Chan's parallel-variance update
//      in the forward pass), derives a / b (or the two backward means) and updates the running statistics;
//   3. apply: same geometry as 1, elementwise, optionally writing a bf16 copy for the convolution that reads it next.
//
// Statistics over several GPUs (one process per GPU, each holding a shard of the batch): the forward pass is split into
// "statistics of my shard" (launches 1-2), an exchange of (mean, M2, count) done by the caller, and the evaluation-mode
// call with the merged statistics (launch 3); the backward pass into "sums of my shard" (launches 1-2), an exchange of
// the two sums, and the apply launch with the merged means.
//
// Conditional mode (reference BigGAN/model.py:61-77: gamma[n, c] * BatchNorm(x) + beta[n, c] with per-sample gain and
// bias from a Linear layer): gamma / beta are [N, C]; slabs never cross a sample, so a slab has one gain per channel.
// Backward: with g = gy * act', ggamma[n, c] = sum_hw g * xh and gbeta[n, c] = sum_hw g per sample, and
//   gx = r * (gamma[n, c] * g - A / P - xh * B / P),  A = sum_n gamma[n, c] * gbeta[n, c],  B = sum_n gamma[n, c] * ggamma[n, c].
#include "common.h"
#include "conv_wgrad.h"

namespace gc {

struct BnParams {          // x / gy / out are fp32 or bf16 (template parameter), all of one type; the rest is fp32
  const void* x;
  const void* gy;
  void* out;             // y | gx
  uint16_t* out_bf16;    // optional bf16 copy of out
  const float* gamma;    // may be null (1)
  const float* beta;     // may be null (0)
  float* running_mean;   // may be null
  float* running_var;
  int64_t* num_batches;  // may be null
  float* save_mean;      // [C]
  float* save_rstd;      // [C]
  float* save_var;       // [C] unbiased batch variance, may be null
  float* ggamma;         // [C]
  float* gbeta;          // [C]
  float* part;           // [slabs][C][2] partial statistics
  float* coef;           // [2][C]  forward: a, b;  backward: gbeta/P, ggamma/P
  int32_t P, C, slabs, slab_pix;
  float eps, momentum, slope;
  int32_t act, training;
  const float* cgamma;   // conditional mode: [N][C] gain and bias (gamma / beta above are then null)
  const float* cbeta;
  int32_t HW, sps, nsamp;   // pixels per sample, slabs per sample (0 = unconditional), samples
};

// pixel range [p0, p1) of a slab; in conditional mode also its sample
__device__ __forceinline__ void slab_range(const BnParams& p, int slab, int& p0, int& p1, int& n) {
  if (p.sps > 0) {
    n = slab / p.sps;
    p0 = n * p.HW + (slab - n * p.sps) * p.slab_pix;
    p1 = min((n + 1) * p.HW, p0 + p.slab_pix);
  } else {
    n = 0;
    p0 = slab * p.slab_pix;
    p1 = min(p.P, p0 + p.slab_pix);
  }
}

constexpr int kQB = 16, kPL = 16;   // channel quads x pixel lanes of a workgroup

__device__ __forceinline__ f32x4 bn_block_sum(f32x4 v, f32x4* red, int tq, int tl) {
  red[tl * kQB + tq] = v;
  __syncthreads();
  if (tl == 0) {
    f32x4 s = v;
    for (int j = 1; j < kPL; j++) s += red[j * kQB + tq];
    red[tq] = s;
  }
  __syncthreads();
  const f32x4 r = red[tq];
  __syncthreads();
  return r;
}

__device__ __forceinline__ f32x4 act_mask(f32x4 pre, int act, float slope) {
  f32x4 m;
#pragma unroll
  for (int e = 0; e < 4; e++) m[e] = (act == GC_ACT_NONE || pre[e] > 0.f) ? 1.f : (act == GC_ACT_RELU ? 0.f : slope);
  return m;
}

// MODE 0: forward slab statistics (mean, M2);  MODE 1: backward slab sums (sum g, sum g * xh)
template <int MODE, typename T>
__global__ __launch_bounds__(kThreads) void bn_partial_kernel(const BnParams p) {
  __shared__ f32x4 red[kThreads];
  const int t = threadIdx.x, tq = t & (kQB - 1), tl = t >> 4;
  const int chunks = (p.C / 4 + kQB - 1) / kQB;
  const int slab = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = (chunk * kQB + tq) * 4;
  const bool cok = c < p.C;
  int p0, p1, n;
  slab_range(p, slab, p0, p1, n);
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto ld = [&](const void* ptr, int pix) -> f32x4 { return load4<T>(ptr, (size_t)pix * p.C + c); };
  if (MODE == 0) {
    f32x4 s = zero;
    if (cok)
      for (int pix = p0 + tl; pix < p1; pix += kPL) s += ld(p.x, pix);
    const f32x4 mean = bn_block_sum(s, red, tq, tl) * (1.f / (float)(p1 - p0));
    f32x4 s2 = zero;
    if (cok)
      for (int pix = p0 + tl; pix < p1; pix += kPL) {
        const f32x4 d = ld(p.x, pix) - mean;
        s2 += d * d;
      }
    s2 = bn_block_sum(s2, red, tq, tl);
    if (cok && tl == 0) {
      *reinterpret_cast<f32x4*>(p.part + ((size_t)slab * p.C + c) * 2) = f32x4{mean[0], s2[0], mean[1], s2[1]};
      *reinterpret_cast<f32x4*>(p.part + ((size_t)slab * p.C + c) * 2 + 4) = f32x4{mean[2], s2[2], mean[3], s2[3]};
    }
  } else {
    f32x4 sg = zero, sgx = zero;
    if (cok) {
      const f32x4 mu = *reinterpret_cast<const f32x4*>(p.save_mean + c), r = *reinterpret_cast<const f32x4*>(p.save_rstd + c);
      f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = zero;
      if (p.gamma != nullptr) ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
      if (p.beta != nullptr) be = *reinterpret_cast<const f32x4*>(p.beta + c);
      if (p.sps > 0) {
        ga = *reinterpret_cast<const f32x4*>(p.cgamma + (size_t)n * p.C + c);
        be = *reinterpret_cast<const f32x4*>(p.cbeta + (size_t)n * p.C + c);
      }
      for (int pix = p0 + tl; pix < p1; pix += kPL) {
        const f32x4 xh = (ld(p.x, pix) - mu) * r;
        const f32x4 g = ld(p.gy, pix) * act_mask(ga * xh + be, p.act, p.slope);
        sg += g;
        sgx += g * xh;
      }
    }
    sg = bn_block_sum(sg, red, tq, tl);
    sgx = bn_block_sum(sgx, red, tq, tl);
    if (cok && tl == 0) {
      *reinterpret_cast<f32x4*>(p.part + ((size_t)slab * p.C + c) * 2) = f32x4{sg[0], sgx[0], sg[1], sgx[1]};
      *reinterpret_cast<f32x4*>(p.part + ((size_t)slab * p.C + c) * 2 + 4) = f32x4{sg[2], sgx[2], sg[3], sgx[3]};
    }
  }
}

// One workgroup per 16 channels: thread (channel, lane) merges slabs lane, lane+16, ... then lane 0 merges the 16 lanes
// (a fixed order, so the result does not depend on scheduling).
constexpr int kFinCh = 16, kFinLanes = kThreads / kFinCh;
constexpr int kDirectRows = 512;      // statistics rows of a convolution up to which the finalizing kernel reads them directly
template <int MODE>
__global__ __launch_bounds__(kThreads) void bn_finalize_kernel(const BnParams p) {
  __shared__ float sh_a[kThreads], sh_b[kThreads], sh_n[kThreads];
  const int t = threadIdx.x, tc = t & (kFinCh - 1), lane = t / kFinCh;
  const int c = blockIdx.x * kFinCh + tc;
  const bool cok = c < p.C;
  float n = 0.f, a = 0.f, b = 0.f;   // forward: count, mean, M2;  backward: -, sum g, sum g*xh
  if (MODE == 2 || MODE == 3) {      // (sum, sum of squares) per slice of the rows the convolution in front wrote: plain sums in double
    __shared__ double sh_s[kThreads], sh_q[kThreads];      // MODE 3: straight from the fp32 rows themselves (few rows: no slice pass)
    double ds = 0.0, dq = 0.0;
    if (cok) {
      int s = lane;
      if (MODE == 3)      // four rows in flight per step (the loads of a row wait for nothing but each other; same fixed order every run)
        for (; s + 3 * kFinLanes < p.slabs; s += 4 * kFinLanes) {
          const float2 v0 = *reinterpret_cast<const float2*>(p.part + ((size_t)s * p.C + c) * 2);
          const float2 v1 = *reinterpret_cast<const float2*>(p.part + ((size_t)(s + kFinLanes) * p.C + c) * 2);
          const float2 v2 = *reinterpret_cast<const float2*>(p.part + ((size_t)(s + 2 * kFinLanes) * p.C + c) * 2);
          const float2 v3 = *reinterpret_cast<const float2*>(p.part + ((size_t)(s + 3 * kFinLanes) * p.C + c) * 2);
          ds += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
          dq += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
        }
      for (; s < p.slabs; s += kFinLanes) {
        if (MODE == 3) {
          const float2 v = *reinterpret_cast<const float2*>(p.part + ((size_t)s * p.C + c) * 2);
          ds += (double)v.x;
          dq += (double)v.y;
        } else {
          const double* v = reinterpret_cast<const double*>(p.part) + ((size_t)s * p.C + c) * 2;
          ds += v[0];
          dq += v[1];
        }
      }
    }
    sh_s[t] = ds; sh_q[t] = dq;
    __syncthreads();
    if (lane != 0 || !cok) return;
    for (int l = 1; l < kFinLanes; l++) { ds += sh_s[l * kFinCh + tc]; dq += sh_q[l * kFinCh + tc]; }
    const double dmean = ds / (double)p.P;
    double m2 = dq - ds * dmean;
    if (m2 < 0.0) m2 = 0.0;
    a = (float)dmean; b = (float)m2;
  } else if (MODE == 1 && p.sps > 0) {      // conditional: per-sample sums are outputs, their gain-weighted totals feed gx
    if (cok)
      for (int smp = lane; smp < p.nsamp; smp += kFinLanes) {
        float sg = 0.f, sgx = 0.f;
        for (int s = 0; s < p.sps; s++) {
          const float2 v = *reinterpret_cast<const float2*>(p.part + ((size_t)(smp * p.sps + s) * p.C + c) * 2);
          sg += v.x;
          sgx += v.y;
        }
        p.gbeta[(size_t)smp * p.C + c] = sg;
        p.ggamma[(size_t)smp * p.C + c] = sgx;
        const float ga = p.cgamma[(size_t)smp * p.C + c];
        a += ga * sg;
        b += ga * sgx;
      }
  } else if (cok)
    for (int s = lane; s < p.slabs; s += kFinLanes) {
      const float2 v = *reinterpret_cast<const float2*>(p.part + ((size_t)s * p.C + c) * 2);
      if (MODE == 0) {
        int q0, q1, smp;
        slab_range(p, s, q0, q1, smp);
        const float nb = (float)(q1 - q0);
        const float tot = n + nb, delta = v.x - a;
        a += delta * (nb / tot);
        b += v.y + delta * delta * (n * nb / tot);
        n = tot;
      } else {
        a += v.x;
        b += v.y;
      }
    }
  if (MODE < 2) {
  sh_a[t] = a; sh_b[t] = b; sh_n[t] = n;
  __syncthreads();
  if (lane != 0 || !cok) return;
  }
  for (int l = 1; MODE < 2 && l < kFinLanes; l++) {
    const float va = sh_a[l * kFinCh + tc], vb = sh_b[l * kFinCh + tc], nb = sh_n[l * kFinCh + tc];
    if (MODE == 0) {
      if (nb > 0.f) {
        const float tot = n + nb, delta = va - a;
        a += delta * (nb / tot);
        b += vb + delta * delta * (n * nb / tot);
        n = tot;
      }
    } else {
      a += va;
      b += vb;
    }
  }
  if (MODE == 0 || MODE >= 2) {
    const float mean = a, var = b / (float)p.P;
    const float r = rsqrtf(var + p.eps);
    p.save_mean[c] = mean;
    if (p.save_rstd != nullptr) p.save_rstd[c] = r;     // (null in the statistics-only call)
    const float ga = p.gamma != nullptr ? p.gamma[c] : 1.f, be = p.beta != nullptr ? p.beta[c] : 0.f;
    p.coef[c] = ga * r;
    p.coef[p.C + c] = be - mean * ga * r;
    const float unbiased = p.P > 1 ? b / (float)(p.P - 1) : var;
    if (p.save_var != nullptr) p.save_var[c] = unbiased;
    if (p.running_mean != nullptr) {
      p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
      p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * unbiased;
    }
    if (p.num_batches != nullptr && c == 0) p.num_batches[0] += 1;
  } else {
    if (p.sps == 0) {
      p.gbeta[c] = a;
      p.ggamma[c] = b;
    }
    p.coef[c] = a / (float)p.P;
    p.coef[p.C + c] = b / (float)p.P;
  }
}

// Rows of (sum, sum of squares) per channel, written by a convolution's epilogue -> one pair of doubles per slice of rows.
// grid: (channel chunks of 16, slices); thread (channel, lane) adds the rows lane, lane + 16, ... of its slice, lane 0
// merges the lanes: a fixed order.
__global__ __launch_bounds__(kThreads) void bn_rows_partial_kernel(const float* rows, double* out, int n_rows, int rows_per_slice, int C) {
  __shared__ double sh_s[kThreads], sh_q[kThreads];
  const int t = threadIdx.x, tc = t & (kFinCh - 1), lane = t / kFinCh;
  const int c = blockIdx.x * kFinCh + tc;
  const int r0 = blockIdx.y * rows_per_slice, r1 = min(n_rows, r0 + rows_per_slice);
  double ds = 0.0, dq = 0.0;
  if (c < C)
    for (int r = r0 + lane; r < r1; r += kFinLanes) {
      const float2 v = *reinterpret_cast<const float2*>(rows + ((size_t)r * C + c) * 2);
      ds += (double)v.x;
      dq += (double)v.y;
    }
  sh_s[t] = ds; sh_q[t] = dq;
  __syncthreads();
  if (lane != 0 || c >= C) return;
  for (int l = 1; l < kFinLanes; l++) { ds += sh_s[l * kFinCh + tc]; dq += sh_q[l * kFinCh + tc]; }
  out[((size_t)blockIdx.y * C + c) * 2] = ds;
  out[((size_t)blockIdx.y * C + c) * 2 + 1] = dq;
}

__global__ __launch_bounds__(kThreads) void bn_update_running_kernel(float* rm, float* rv, int64_t* nbt, const float* mean, const float* var,
                                                                    int C, float momentum) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c == 0 && nbt != nullptr) nbt[0] += 1;
  if (c >= C) return;
  rm[c] = (1.f - momentum) * rm[c] + momentum * mean[c];
  rv[c] = (1.f - momentum) * rv[c] + momentum * var[c];
}

// Evaluation mode: a / b from the running statistics.
__global__ __launch_bounds__(kThreads) void bn_eval_coef_kernel(const BnParams p) {
  const int c = blockIdx.x * kThreads + threadIdx.x;
  if (c >= p.C) return;
  const float r = rsqrtf(p.running_var[c] + p.eps), mean = p.running_mean[c];
  const float ga = p.gamma != nullptr ? p.gamma[c] : 1.f, be = p.beta != nullptr ? p.beta[c] : 0.f;
  p.save_mean[c] = mean;
  p.save_rstd[c] = r;
  p.coef[c] = ga * r;
  p.coef[p.C + c] = be - mean * ga * r;
}

// MODE 0: y = act(a x + b);  MODE 1: gx = gamma r (g - gbeta/P - xh ggamma/P)
template <int MODE, typename T>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const BnParams p) {
  const int t = threadIdx.x, tq = t & (kQB - 1), tl = t >> 4;
  const int chunks = (p.C / 4 + kQB - 1) / kQB;
  const int slab = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = (chunk * kQB + tq) * 4;
  if (c >= p.C) return;
  int p0, p1, n;
  slab_range(p, slab, p0, p1, n);
  const f32x4 k0 = *reinterpret_cast<const f32x4*>(p.coef + c), k1 = *reinterpret_cast<const f32x4*>(p.coef + p.C + c);
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, r = mu, ga = {1.f, 1.f, 1.f, 1.f}, be = mu;
  if (MODE == 1) {
    mu = *reinterpret_cast<const f32x4*>(p.save_mean + c);
    r = *reinterpret_cast<const f32x4*>(p.save_rstd + c);
    if (p.gamma != nullptr) ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
    if (p.beta != nullptr) be = *reinterpret_cast<const f32x4*>(p.beta + c);
  }
  const bool cond = p.sps > 0;
  if (cond) {
    ga = *reinterpret_cast<const f32x4*>(p.cgamma + (size_t)n * p.C + c);
    be = *reinterpret_cast<const f32x4*>(p.cbeta + (size_t)n * p.C + c);
  }
  for (int pix = p0 + tl; pix < p1; pix += kPL) {
    const size_t o = (size_t)pix * p.C + c;
    const f32x4 xv = load4<T>(p.x, o);
    f32x4 res;
    if (MODE == 0) {
      f32x4 pre = k0 * xv + k1;
      if (cond) pre = ga * pre + be;      // k0, k1 normalise only in conditional mode
      const f32x4 m = act_mask(pre, p.act, p.slope);
      res = pre * m;
    } else {
      const f32x4 xh = (xv - mu) * r;
      const f32x4 g = load4<T>(p.gy, o) * act_mask(ga * xh + be, p.act, p.slope);
      res = cond ? r * (ga * g - k0 - xh * k1) : ga * r * (g - k0 - xh * k1);
    }
    store4<T>(p.out, o, res);
    if (p.out_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out_bf16 + o) = cvt_bf16x4(res);
  }
}

// Small tensors (at most kSmallPBn pixels): one launch per direction.  A workgroup owns 16 channels (4 quads x 64 pixel
// lanes) and all pixels: sum, centred squares (from L1 / L2) and the apply sweep, the running statistics included -- the
// three-launch form costs 25-30 us on such tensors for a few us of work.
constexpr int kSmallP = 4096, kSmallQB = 4, kSmallPL = kThreads / kSmallQB;
// (BatchNorm takes that form up to 1024 pixels only: its sweeps are chains of dependent loads, 16 per lane at that size;
//  at 4096 pixels x 512 channels the single launch measured 48 us against 33 us for the three launches)
constexpr int kSmallPBn = 1024;

__device__ __forceinline__ f32x4 bn_small_sum(f32x4 v, f32x4* red, int tq, int tl) {
  red[tl * kSmallQB + tq] = v;
  __syncthreads();
  if (tl < 8) {                       // 64 lanes -> 8 -> 1
    f32x4 s = red[tl * kSmallQB + tq];
    for (int j = tl + 8; j < kSmallPL; j += 8) s += red[j * kSmallQB + tq];
    red[tl * kSmallQB + tq] = s;      // (each (tl, tq) slot is read and written by its own thread only in this step)
  }
  __syncthreads();
  f32x4 r = red[tq];
  for (int j = 1; j < 8; j++) r += red[j * kSmallQB + tq];
  __syncthreads();
  return r;
}

template <int MODE, typename T>
__global__ __launch_bounds__(kThreads) void bn_small_kernel(const BnParams p) {
  __shared__ f32x4 red[kThreads];
  const int t = threadIdx.x, tq = t & (kSmallQB - 1), tl = t / kSmallQB;
  const int c = (blockIdx.x * kSmallQB + tq) * 4;
  const bool cok = c < p.C;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto ld = [&](const void* ptr, int pix) -> f32x4 { return load4<T>(ptr, (size_t)pix * p.C + c); };
  f32x4 ga = {1.f, 1.f, 1.f, 1.f}, be = zero;
  if (cok && p.gamma != nullptr) ga = *reinterpret_cast<const f32x4*>(p.gamma + c);
  if (cok && p.beta != nullptr) be = *reinterpret_cast<const f32x4*>(p.beta + c);
  const float inv = 1.f / (float)p.P;
  if (MODE == 0) {
    f32x4 s = zero;
    if (cok)
      for (int pix = tl; pix < p.P; pix += kSmallPL) s += ld(p.x, pix);
    const f32x4 mean = bn_small_sum(s, red, tq, tl) * inv;
    f32x4 s2 = zero;
    if (cok)
      for (int pix = tl; pix < p.P; pix += kSmallPL) {
        const f32x4 d = ld(p.x, pix) - mean;
        s2 += d * d;
      }
    s2 = bn_small_sum(s2, red, tq, tl);
    if (!cok) return;
    f32x4 r, unbiased;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      r[e] = rsqrtf(s2[e] * inv + p.eps);
      unbiased[e] = p.P > 1 ? s2[e] / (float)(p.P - 1) : s2[e] * inv;
    }
    if (tl == 0) {
      *reinterpret_cast<f32x4*>(p.save_mean + c) = mean;
      *reinterpret_cast<f32x4*>(p.save_rstd + c) = r;
      if (p.save_var != nullptr) *reinterpret_cast<f32x4*>(p.save_var + c) = unbiased;
      if (p.running_mean != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          p.running_mean[c + e] = (1.f - p.momentum) * p.running_mean[c + e] + p.momentum * mean[e];
          p.running_var[c + e] = (1.f - p.momentum) * p.running_var[c + e] + p.momentum * unbiased[e];
        }
      }
      if (p.num_batches != nullptr && c == 0) p.num_batches[0] += 1;
    }
    const f32x4 k0 = ga * r, k1 = be - mean * ga * r;
    for (int pix = tl; pix < p.P; pix += kSmallPL) {
      const size_t o = (size_t)pix * p.C + c;
      const f32x4 pre = k0 * load4<T>(p.x, o) + k1;
      const f32x4 res = pre * act_mask(pre, p.act, p.slope);
      store4<T>(p.out, o, res);
      if (p.out_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out_bf16 + o) = cvt_bf16x4(res);
    }
  } else {
    const f32x4 mu = cok ? *reinterpret_cast<const f32x4*>(p.save_mean + c) : zero, r = cok ? *reinterpret_cast<const f32x4*>(p.save_rstd + c) : zero;
    f32x4 sg = zero, sgx = zero;
    if (cok)
      for (int pix = tl; pix < p.P; pix += kSmallPL) {
        const f32x4 xh = (ld(p.x, pix) - mu) * r;
        const f32x4 g = ld(p.gy, pix) * act_mask(ga * xh + be, p.act, p.slope);
        sg += g;
        sgx += g * xh;
      }
    sg = bn_small_sum(sg, red, tq, tl);
    sgx = bn_small_sum(sgx, red, tq, tl);
    if (!cok) return;
    if (tl == 0) {
      *reinterpret_cast<f32x4*>(p.gbeta + c) = sg;
      *reinterpret_cast<f32x4*>(p.ggamma + c) = sgx;
    }
    const f32x4 k0 = sg * inv, k1 = sgx * inv;
    for (int pix = tl; pix < p.P; pix += kSmallPL) {
      const size_t o = (size_t)pix * p.C + c;
      const f32x4 xh = (load4<T>(p.x, o) - mu) * r;
      const f32x4 g = load4<T>(p.gy, o) * act_mask(ga * xh + be, p.act, p.slope);
      const f32x4 res = ga * r * (g - k0 - xh * k1);
      store4<T>(p.out, o, res);
      if (p.out_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out_bf16 + o) = cvt_bf16x4(res);
    }
  }
}

// Per-channel sums over all pixels (the bias gradient of a convolution: reference torch.nn.Conv2d(..., bias=True) at
// Pix2Pix/models.py:17,225-232, StyleGAN/model.py:72): slab sums with the geometry of the statistics kernels, then one
// small launch that adds the slabs in a fixed order (bitwise reproducible, no atomics).
template <typename T>
__global__ __launch_bounds__(kThreads) void channel_sum_partial_kernel(const BnParams p) {
  __shared__ f32x4 red[kThreads];
  const int t = threadIdx.x, tq = t & (kQB - 1), tl = t >> 4;
  const int chunks = (p.C / 4 + kQB - 1) / kQB;
  const int slab = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = (chunk * kQB + tq) * 4;
  const bool cok = c < p.C;
  int p0, p1, n;
  slab_range(p, slab, p0, p1, n);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    int pix = p0 + tl;
    for (; pix + 3 * kPL < p1; pix += 4 * kPL) {       // four independent loads in flight
      const f32x4 a = load4<T>(p.x, (size_t)pix * p.C + c), b = load4<T>(p.x, (size_t)(pix + kPL) * p.C + c);
      const f32x4 d = load4<T>(p.x, (size_t)(pix + 2 * kPL) * p.C + c), e = load4<T>(p.x, (size_t)(pix + 3 * kPL) * p.C + c);
      s += (a + b) + (d + e);
    }
    for (; pix < p1; pix += kPL) s += load4<T>(p.x, (size_t)pix * p.C + c);
  }
  s = bn_block_sum(s, red, tq, tl);
  if (cok && tl == 0) *reinterpret_cast<f32x4*>(p.part + (size_t)slab * p.C + c) = s;
}

// The gradient through an activation that was fused into a convolution's epilogue, from the activation's OUTPUT y, and
// the bias gradient of that convolution in the same sweep:  out = gy * act'(y),  slab sums of out (added by
// channel_sum_final_kernel).  act'(y): ReLU y > 0; LeakyReLU y > 0 ? 1 : slope; tanh 1 - y^2; sigmoid y (1 - y).
template <typename T>
__global__ __launch_bounds__(kThreads) void act_backward_sum_kernel(const BnParams p) {
  __shared__ f32x4 red[kThreads];
  const int t = threadIdx.x, tq = t & (kQB - 1), tl = t >> 4;
  const int chunks = (p.C / 4 + kQB - 1) / kQB;
  const int slab = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = (chunk * kQB + tq) * 4;
  const bool cok = c < p.C;
  int p0, p1, n;
  slab_range(p, slab, p0, p1, n);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](size_t o, f32x4 y, f32x4 g) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float d = p.act == GC_ACT_RELU ? (y[e] > 0.f ? 1.f : 0.f)
                    : p.act == GC_ACT_LRELU ? (y[e] > 0.f ? 1.f : p.slope)
                    : p.act == GC_ACT_TANH ? 1.f - y[e] * y[e] : y[e] * (1.f - y[e]);
      r[e] = g[e] * d;
    }
    store4<T>(p.out, o, r);
    if (sizeof(T) == 2) r = widen_bf16x4(cvt_bf16x4(r));      // the sum is that of the stored (rounded) values
    s += r;
  };
  if (cok) {
    int pix = p0 + tl;
    for (; pix + 3 * kPL < p1; pix += 4 * kPL) {         // eight independent loads in flight
      size_t o[4];
      f32x4 y[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        o[u] = (size_t)(pix + u * kPL) * p.C + c;
        y[u] = load4<T>(p.x, o[u]);
        g[u] = load4<T>(p.gy, o[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) one(o[u], y[u], g[u]);
    }
    for (; pix < p1; pix += kPL) {
      const size_t o = (size_t)pix * p.C + c;
      one(o, load4<T>(p.x, o), load4<T>(p.gy, o));
    }
  }
  s = bn_block_sum(s, red, tq, tl);
  if (cok && tl == 0) *reinterpret_cast<f32x4*>(p.part + (size_t)slab * p.C + c) = s;
}

// 16 channels x 16 slab lanes per workgroup: lane l adds slabs l, l + 16, ... (independent loads), then the 16 lanes are
// added in a fixed order.
__global__ __launch_bounds__(kThreads) void channel_sum_final_kernel(const float* part, float* out, int slabs, int C) {
  __shared__ float red[kThreads];
  const int t = threadIdx.x, tc = t & 15, lane = t >> 4;
  const int c = blockIdx.x * 16 + tc;
  float s = 0.f;
  if (c < C)
    for (int k = lane; k < slabs; k += 16) s += part[(size_t)k * C + c];
  red[t] = s;
  __syncthreads();
  if (lane == 0 && c < C) {
    for (int l = 1; l < 16; l++) s += red[l * 16 + tc];
    out[c] = s;
  }
}

// Few pixels: one launch, a workgroup owns 16 channels and all pixels (same summation tree for every launch).
template <typename T>
__global__ __launch_bounds__(kThreads) void channel_sum_small_kernel(const void* x, float* out, int P, int C) {
  __shared__ f32x4 red[kThreads];
  const int t = threadIdx.x, tq = t & (kSmallQB - 1), tl = t / kSmallQB;
  const int c = (blockIdx.x * kSmallQB + tq) * 4;
  const bool cok = c < C;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    int pix = tl;
    for (; pix + 3 * kSmallPL < P; pix += 4 * kSmallPL) {      // four independent loads in flight
      const f32x4 a = load4<T>(x, (size_t)pix * C + c), b = load4<T>(x, (size_t)(pix + kSmallPL) * C + c);
      const f32x4 d = load4<T>(x, (size_t)(pix + 2 * kSmallPL) * C + c), e = load4<T>(x, (size_t)(pix + 3 * kSmallPL) * C + c);
      s += (a + b) + (d + e);
    }
    for (; pix < P; pix += kSmallPL) s += load4<T>(x, (size_t)pix * C + c);
  }
  s = bn_small_sum(s, red, tq, tl);
  if (cok && tl == 0) *reinterpret_cast<f32x4*>(out + c) = s;
}

static void bn_geometry(BnParams& p) {
  if (p.nsamp > 0) {   // conditional: slabs inside samples, about 2048 in all, at least 256 pixels each
    long sps = 2048 / p.nsamp;
    if (sps > p.HW / 256) sps = p.HW / 256;
    if (sps < 1) sps = 1;
    p.slab_pix = (int)((p.HW + sps - 1) / sps);
    p.sps = (p.HW + p.slab_pix - 1) / p.slab_pix;
    p.slabs = p.nsamp * p.sps;
    return;
  }
  const int chunks = (p.C / 4 + kQB - 1) / kQB;
  long slabs = (p.P + 255) / 256;              // at least 256 pixels per slab
  const long cap = chunks >= 4 ? 128 : 512 / chunks;
  if (slabs > cap) slabs = cap;
  if (slabs < 1) slabs = 1;
  p.slab_pix = (int)((p.P + slabs - 1) / slabs);
  p.slabs = (p.P + p.slab_pix - 1) / p.slab_pix;
}

static bool misaligned(const void* q, unsigned mask) { return (reinterpret_cast<uintptr_t>(q) & mask) != 0; }

}  // namespace gc

using namespace gc;

extern "C" {

static size_t bn_workspace(int64_t P, int32_t C, int32_t nsamp) {
  if (P <= 0 || C <= 0 || P >= (1ll << 31)) return 0;
  BnParams p{};
  p.P = (int32_t)P; p.C = C; p.nsamp = nsamp; p.HW = nsamp > 0 ? (int32_t)(P / nsamp) : 0;
  bn_geometry(p);
  return ((size_t)p.slabs * C * 2 + 2 * (size_t)C) * sizeof(float);
}

size_t gc_batchnorm_workspace_bytes(int64_t P, int32_t C) { return bn_workspace(P, C, 0); }

size_t gc_cond_batchnorm_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
  if (N <= 0 || HW <= 0) return 0;
  return bn_workspace((int64_t)N * HW, C, N);
}

static int bn_forward(const void* x, void* y, void* y_bf16, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                      void* workspace, size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, float eps, float momentum,
                      int32_t act, float slope, int32_t training, void* stream, int32_t nsamp, const float* stat_rows = nullptr,
                      int32_t n_rows = 0) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("batchnorm: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (dtype == GC_BF16 && y_bf16 != nullptr) { set_error("batchnorm: bf16 tensors need no bf16 copy"); return GC_BAD_ARGUMENT; }
  if (P < 0 || C <= 0 || P >= (1ll << 31)) { set_error("batchnorm: bad shape"); return GC_BAD_ARGUMENT; }
  if (act != GC_ACT_NONE && act != GC_ACT_RELU && act != GC_ACT_LRELU) { set_error("batchnorm: activation must be none, ReLU or LeakyReLU"); return GC_BAD_ARGUMENT; }
  if (C % 4 != 0) { set_error("batchnorm: channels must be a multiple of 4 (C=%d)", C); return GC_UNSUPPORTED_GEOMETRY; }
  if (P == 0) return GC_OK;
  if (x == nullptr || y == nullptr || save_mean == nullptr || save_rstd == nullptr || workspace == nullptr) { set_error("batchnorm: null pointer"); return GC_BAD_ARGUMENT; }
  if (!training && (running_mean == nullptr || running_var == nullptr)) { set_error("batchnorm: evaluation mode needs running statistics"); return GC_BAD_ARGUMENT; }
  if ((running_mean == nullptr) != (running_var == nullptr)) { set_error("batchnorm: running_mean and running_var go together"); return GC_BAD_ARGUMENT; }
  const unsigned am = dtype == GC_F32 ? 15 : 7;
  if (misaligned(x, am) || misaligned(y, am) || misaligned(y_bf16, 7) || misaligned(workspace, 15) || misaligned(gamma, 15) ||
      misaligned(beta, 15) || misaligned(save_mean, 15) || misaligned(save_rstd, 15)) {
    set_error("batchnorm: misaligned pointer"); return GC_BAD_ARGUMENT;
  }
  if (nsamp > 0 && (gamma == nullptr || beta == nullptr)) { set_error("batchnorm: conditional mode needs gain and bias"); return GC_BAD_ARGUMENT; }
  const size_t need = bn_workspace(P, C, nsamp);
  if (workspace_bytes < need) { set_error("batchnorm: workspace of %zu bytes given, %zu needed", workspace_bytes, need); return GC_WORKSPACE_TOO_SMALL; }
  BnParams p{};
  p.x = x; p.out = y; p.out_bf16 = (uint16_t*)y_bf16;
  if (nsamp > 0) { p.cgamma = gamma; p.cbeta = beta; p.nsamp = nsamp; p.HW = (int32_t)(P / nsamp); }
  else { p.gamma = gamma; p.beta = beta; }
  p.running_mean = running_mean; p.running_var = running_var; p.num_batches = num_batches_tracked;
  p.save_mean = save_mean; p.save_rstd = save_rstd; p.save_var = save_var;
  p.P = (int32_t)P; p.C = C; p.eps = eps; p.momentum = momentum; p.slope = slope; p.act = act; p.training = training;
  bn_geometry(p);
  p.part = (float*)workspace;
  p.coef = p.part + (size_t)p.slabs * C * 2;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (C / 4 + kQB - 1) / kQB;
  const dim3 grid((unsigned)(p.slabs * chunks)), block(kThreads);
  if (stat_rows != nullptr) {          // statistics from the rows the convolution in front wrote: finalize + apply
    int slices = n_rows / 128;
    if (slices < 1) slices = 1;
    if (slices > 64) slices = 64;
    const int per = (n_rows + slices - 1) / slices;
    slices = (n_rows + per - 1) / per;
    if ((size_t)slices * C * 2 * sizeof(double) > (size_t)p.slabs * C * 2 * sizeof(float)) {
      set_error("batchnorm: workspace too small for %d slices of statistics rows", slices); return GC_WORKSPACE_TOO_SMALL;
    }
    BnParams q = p;
    q.slabs = slices;
    if (n_rows <= kDirectRows) {       // few rows: the finalizing workgroups add them themselves (one launch less)
      q.slabs = n_rows;
      q.part = const_cast<float*>(stat_rows);
      ProfScope ps("gc::bn_finalize_kernel<3>", 0.0, s);
      hipLaunchKernelGGL(bn_finalize_kernel<3>, dim3((C + kFinCh - 1) / kFinCh), block, 0, s, q);
    } else {
      {
        ProfScope ps("gc::bn_rows_partial_kernel", 0.0, s);
        hipLaunchKernelGGL(bn_rows_partial_kernel, dim3((C + kFinCh - 1) / kFinCh, slices), block, 0, s, stat_rows, reinterpret_cast<double*>(p.part),
                           n_rows, per, C);
      }
      { ProfScope ps("gc::bn_finalize_kernel<2>", 0.0, s); hipLaunchKernelGGL(bn_finalize_kernel<2>, dim3((C + kFinCh - 1) / kFinCh), block, 0, s, q); }
    }
    ProfScope ps("gc::bn_apply_kernel<0>", 0.0, s);
    if (dtype == GC_F32) hipLaunchKernelGGL((bn_apply_kernel<0, float>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((bn_apply_kernel<0, uint16_t>), grid, block, 0, s, p);
    return check_hip(hipGetLastError(), "batchnorm forward launch");
  }
  if (training && nsamp == 0 && P <= kSmallPBn) {
    ProfScope ps("gc::bn_small_kernel<0>", 0.0, s);
    const dim3 sgrid((unsigned)((C / 4 + kSmallQB - 1) / kSmallQB));
    if (dtype == GC_F32) hipLaunchKernelGGL((bn_small_kernel<0, float>), sgrid, block, 0, s, p);
    else hipLaunchKernelGGL((bn_small_kernel<0, uint16_t>), sgrid, block, 0, s, p);
    return check_hip(hipGetLastError(), "batchnorm forward launch");
  }
  if (training) {
    {
      ProfScope ps("gc::bn_partial_kernel<0>", 0.0, s);
      if (dtype == GC_F32) hipLaunchKernelGGL((bn_partial_kernel<0, float>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((bn_partial_kernel<0, uint16_t>), grid, block, 0, s, p);
    }
    { ProfScope ps("gc::bn_finalize_kernel<0>", 0.0, s); hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3((C + kFinCh - 1) / kFinCh), block, 0, s, p); }
  } else {
    ProfScope ps("gc::bn_eval_coef_kernel", 0.0, s);
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + kThreads - 1) / kThreads), block, 0, s, p);
  }
  {
    ProfScope ps("gc::bn_apply_kernel<0>", 0.0, s);
    if (dtype == GC_F32) hipLaunchKernelGGL((bn_apply_kernel<0, float>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((bn_apply_kernel<0, uint16_t>), grid, block, 0, s, p);
  }
  return check_hip(hipGetLastError(), "batchnorm forward launch");
}

int gc_batchnorm_act_fwd(const void* x, void* y, void* y_bf16, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                         void* workspace, size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, float eps, float momentum,
                         int32_t act, float slope, int32_t training, void* stream) {
  return bn_forward(x, y, y_bf16, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, save_var, workspace,
                    workspace_bytes, P, C, dtype, eps, momentum, act, slope, training, stream, 0);
}

int gc_batchnorm_act_fwd_rows(const void* x, void* y, void* y_bf16, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                              const float* stat_rows, int32_t n_rows, void* workspace, size_t workspace_bytes, int64_t P, int32_t C,
                              int32_t dtype, float eps, float momentum, int32_t act, float slope, void* stream) {
  if (stat_rows == nullptr || n_rows <= 0 || misaligned(stat_rows, 7)) { set_error("batchnorm: rows of statistics expected"); return GC_BAD_ARGUMENT; }
  return bn_forward(x, y, y_bf16, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, save_var, workspace,
                    workspace_bytes, P, C, dtype, eps, momentum, act, slope, 1, stream, 0, stat_rows, n_rows);
}

int gc_cond_batchnorm_act_fwd(const void* x, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var, void* workspace,
                              size_t workspace_bytes, int32_t N, int32_t HW, int32_t C, int32_t dtype, float eps, float momentum,
                              int32_t act, float slope, int32_t training, void* stream) {
  if (N <= 0 || HW <= 0) { set_error("batchnorm: conditional mode needs N > 0 and HW > 0"); return GC_BAD_ARGUMENT; }
  return bn_forward(x, y, nullptr, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, save_var, workspace,
                    workspace_bytes, (int64_t)N * HW, C, dtype, eps, momentum, act, slope, training, stream, N);
}

int gc_cond_batchnorm_act_fwd_rows(const void* x, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                                   const float* stat_rows, int32_t n_rows, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW,
                                   int32_t C, int32_t dtype, float eps, float momentum, int32_t act, float slope, void* stream) {
  if (N <= 0 || HW <= 0) { set_error("batchnorm: conditional mode needs N > 0 and HW > 0"); return GC_BAD_ARGUMENT; }
  if (stat_rows == nullptr || n_rows <= 0 || misaligned(stat_rows, 7)) { set_error("batchnorm: rows of statistics expected"); return GC_BAD_ARGUMENT; }
  return bn_forward(x, y, nullptr, gamma, beta, running_mean, running_var, num_batches_tracked, save_mean, save_rstd, save_var, workspace,
                    workspace_bytes, (int64_t)N * HW, C, dtype, eps, momentum, act, slope, 1, stream, N, stat_rows, n_rows);
}

static int bn_backward(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                       const float* save_rstd, void* gx, void* gx_bf16, float* ggamma, float* gbeta, void* workspace,
                       size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, int32_t act, float slope, void* stream, int32_t nsamp,
                       int phase = 0, const float* means = nullptr) {
  // phase 0: everything;  1: the two sums only (gx unused);  2: apply only, with `means` = [2][C] (sum g, sum g*xh) / count
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("batchnorm: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (dtype == GC_BF16 && gx_bf16 != nullptr) { set_error("batchnorm: bf16 tensors need no bf16 copy"); return GC_BAD_ARGUMENT; }
  if (P < 0 || C <= 0 || P >= (1ll << 31)) { set_error("batchnorm: bad shape"); return GC_BAD_ARGUMENT; }
  if (act != GC_ACT_NONE && act != GC_ACT_RELU && act != GC_ACT_LRELU) { set_error("batchnorm: activation must be none, ReLU or LeakyReLU"); return GC_BAD_ARGUMENT; }
  if (C % 4 != 0) { set_error("batchnorm: channels must be a multiple of 4 (C=%d)", C); return GC_UNSUPPORTED_GEOMETRY; }
  if (x == nullptr || gy == nullptr || (gx == nullptr && phase != 1) || ((ggamma == nullptr || gbeta == nullptr) && phase != 2) ||
      save_mean == nullptr || save_rstd == nullptr || (workspace == nullptr && phase != 2) || (means == nullptr && phase == 2)) {
    set_error("batchnorm: null pointer"); return GC_BAD_ARGUMENT;
  }
  const unsigned am = dtype == GC_F32 ? 15 : 7;
  if (misaligned(x, am) || misaligned(gy, am) || misaligned(gx, am) || misaligned(gx_bf16, 7) || misaligned(workspace, 15) ||
      misaligned(gamma, 15) || misaligned(beta, 15) || misaligned(save_mean, 15) || misaligned(save_rstd, 15) || misaligned(means, 15)) {
    set_error("batchnorm: misaligned pointer"); return GC_BAD_ARGUMENT;
  }
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {
    if (phase == 2) return GC_OK;
    if (zero_fill(ggamma, sizeof(float) * C, s) != GC_OK) return GC_HIP_ERROR;
    return zero_fill(gbeta, sizeof(float) * C, s);
  }
  if (nsamp > 0 && (gamma == nullptr || beta == nullptr)) { set_error("batchnorm: conditional mode needs gain and bias"); return GC_BAD_ARGUMENT; }
  const size_t need = phase == 2 ? 0 : bn_workspace(P, C, nsamp);
  if (workspace_bytes < need) { set_error("batchnorm: workspace of %zu bytes given, %zu needed", workspace_bytes, need); return GC_WORKSPACE_TOO_SMALL; }
  BnParams p{};
  p.x = x; p.gy = gy; p.out = gx; p.out_bf16 = (uint16_t*)gx_bf16;
  if (nsamp > 0) { p.cgamma = gamma; p.cbeta = beta; p.nsamp = nsamp; p.HW = (int32_t)(P / nsamp); }
  else { p.gamma = gamma; p.beta = beta; }
  p.save_mean = const_cast<float*>(save_mean); p.save_rstd = const_cast<float*>(save_rstd);
  p.ggamma = ggamma; p.gbeta = gbeta;
  p.P = (int32_t)P; p.C = C; p.slope = slope; p.act = act;
  bn_geometry(p);
  p.part = (float*)workspace;
  p.coef = phase == 2 ? const_cast<float*>(means) : p.part + (size_t)p.slabs * C * 2;
  const int chunks = (C / 4 + kQB - 1) / kQB;
  const dim3 grid((unsigned)(p.slabs * chunks)), block(kThreads);
  if (phase == 0 && nsamp == 0 && P <= kSmallPBn) {
    ProfScope ps("gc::bn_small_kernel<1>", 0.0, s);
    const dim3 sgrid((unsigned)((C / 4 + kSmallQB - 1) / kSmallQB));
    if (dtype == GC_F32) hipLaunchKernelGGL((bn_small_kernel<1, float>), sgrid, block, 0, s, p);
    else hipLaunchKernelGGL((bn_small_kernel<1, uint16_t>), sgrid, block, 0, s, p);
    return check_hip(hipGetLastError(), "batchnorm backward launch");
  }
  if (phase != 2) {
    {
      ProfScope ps("gc::bn_partial_kernel<1>", 0.0, s);
      if (dtype == GC_F32) hipLaunchKernelGGL((bn_partial_kernel<1, float>), grid, block, 0, s, p);
      else hipLaunchKernelGGL((bn_partial_kernel<1, uint16_t>), grid, block, 0, s, p);
    }
    { ProfScope ps("gc::bn_finalize_kernel<1>", 0.0, s); hipLaunchKernelGGL(bn_finalize_kernel<1>, dim3((C + kFinCh - 1) / kFinCh), block, 0, s, p); }
  }
  if (phase != 1) {
    ProfScope ps("gc::bn_apply_kernel<1>", 0.0, s);
    if (dtype == GC_F32) hipLaunchKernelGGL((bn_apply_kernel<1, float>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((bn_apply_kernel<1, uint16_t>), grid, block, 0, s, p);
  }
  return check_hip(hipGetLastError(), "batchnorm backward launch");
}

int gc_batchnorm_act_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_rstd, void* gx, void* gx_bf16, float* ggamma, float* gbeta, void* workspace,
                         size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, int32_t act, float slope, void* stream) {
  return bn_backward(x, gy, gamma, beta, save_mean, save_rstd, gx, gx_bf16, ggamma, gbeta, workspace, workspace_bytes, P, C, dtype, act,
                     slope, stream, 0);
}

int gc_cond_batchnorm_act_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_rstd, void* gx, float* ggamma, float* gbeta, void* workspace, size_t workspace_bytes,
                              int32_t N, int32_t HW, int32_t C, int32_t dtype, int32_t act, float slope, void* stream) {
  if (N <= 0 || HW <= 0) { set_error("batchnorm: conditional mode needs N > 0 and HW > 0"); return GC_BAD_ARGUMENT; }
  return bn_backward(x, gy, gamma, beta, save_mean, save_rstd, gx, nullptr, ggamma, gbeta, workspace, workspace_bytes, (int64_t)N * HW, C,
                     dtype, act, slope, stream, N);
}

size_t gc_channel_sum_workspace_bytes(int64_t P, int32_t C) {
  if (P <= 0 || C <= 0 || P >= (1ll << 31) || C % 4 != 0) return 0;
  BnParams p{};
  p.P = (int32_t)P; p.C = C;
  bn_geometry(p);
  return (size_t)p.slabs * C * sizeof(float);
}

int gc_channel_sum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, void* stream) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("gc_channel_sum: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (P < 0 || C <= 0 || P >= (1ll << 31)) { set_error("gc_channel_sum: bad shape"); return GC_BAD_ARGUMENT; }
  if (out == nullptr || (P > 0 && x == nullptr)) { set_error("gc_channel_sum: null pointer"); return GC_BAD_ARGUMENT; }
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) return zero_fill(out, sizeof(float) * C, s);
  if (C % 4 != 0 || misaligned(x, dtype == GC_F32 ? 15 : 7)) {
    // odd channel counts (the one- and three-channel ends of the networks): element loads, fp32 atomic adds
    ProfScope ps("gc::colsum_kernel", 0.0, s);
    return launch_colsum(x, out, (int)P, C, C, 0, dtype == GC_BF16, s);
  }
  if (P <= kSmallP && !misaligned(out, 15)) {
    ProfScope ps("gc::channel_sum_kernel", 0.0, s);
    const dim3 grid((unsigned)((C / 4 + kSmallQB - 1) / kSmallQB));
    if (dtype == GC_F32) hipLaunchKernelGGL(channel_sum_small_kernel<float>, grid, dim3(kThreads), 0, s, x, out, (int)P, C);
    else hipLaunchKernelGGL(channel_sum_small_kernel<uint16_t>, grid, dim3(kThreads), 0, s, x, out, (int)P, C);
    return check_hip(hipGetLastError(), "channel sum launch");
  }
  const size_t need = gc_channel_sum_workspace_bytes(P, C);
  if (workspace == nullptr || misaligned(workspace, 15) || workspace_bytes < need) {
    set_error("gc_channel_sum: workspace of %zu bytes given, %zu needed (16-byte aligned)", workspace_bytes, need);
    return GC_WORKSPACE_TOO_SMALL;
  }
  BnParams p{};
  p.x = x; p.P = (int32_t)P; p.C = C;
  bn_geometry(p);
  p.part = (float*)workspace;
  const int chunks = (C / 4 + kQB - 1) / kQB;
  ProfScope ps("gc::channel_sum_kernel", 0.0, s);
  if (dtype == GC_F32) hipLaunchKernelGGL(channel_sum_partial_kernel<float>, dim3((unsigned)(p.slabs * chunks)), dim3(kThreads), 0, s, p);
  else hipLaunchKernelGGL(channel_sum_partial_kernel<uint16_t>, dim3((unsigned)(p.slabs * chunks)), dim3(kThreads), 0, s, p);
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3((unsigned)((C + 15) / 16)), dim3(kThreads), 0, s, p.part, out, p.slabs, C);
  return check_hip(hipGetLastError(), "channel sum launch");
}

int gc_act_backward_channel_sum(const void* gy, const void* y, void* out, float* colsum, void* workspace, size_t workspace_bytes,
                                int64_t P, int32_t C, int32_t dtype, int32_t act, float slope, void* stream) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("gc_act_backward_channel_sum: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (act != GC_ACT_RELU && act != GC_ACT_LRELU && act != GC_ACT_TANH && act != GC_ACT_SIGMOID) {
    set_error("gc_act_backward_channel_sum: activation must be ReLU, LeakyReLU, tanh or sigmoid"); return GC_BAD_ARGUMENT;
  }
  if (P < 0 || C <= 0 || P >= (1ll << 31)) { set_error("gc_act_backward_channel_sum: bad shape"); return GC_BAD_ARGUMENT; }
  if (C % 4 != 0) { set_error("gc_act_backward_channel_sum: channels must be a multiple of 4 (C=%d)", C); return GC_UNSUPPORTED_GEOMETRY; }
  if (colsum == nullptr) { set_error("gc_act_backward_channel_sum: null pointer"); return GC_BAD_ARGUMENT; }
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) return zero_fill(colsum, sizeof(float) * C, s);
  const unsigned am = dtype == GC_F32 ? 15 : 7;
  if (gy == nullptr || y == nullptr || out == nullptr || misaligned(gy, am) || misaligned(y, am) || misaligned(out, am)) {
    set_error("gc_act_backward_channel_sum: null or misaligned tensor"); return GC_BAD_ARGUMENT;
  }
  const size_t need = gc_channel_sum_workspace_bytes(P, C);
  if (workspace == nullptr || misaligned(workspace, 15) || workspace_bytes < need) {
    set_error("gc_act_backward_channel_sum: workspace of %zu bytes given, %zu needed (16-byte aligned)", workspace_bytes, need);
    return GC_WORKSPACE_TOO_SMALL;
  }
  BnParams p{};
  p.x = y; p.gy = gy; p.out = out; p.P = (int32_t)P; p.C = C; p.act = act; p.slope = slope;
  bn_geometry(p);
  p.part = (float*)workspace;
  const int chunks = (C / 4 + kQB - 1) / kQB;
  ProfScope ps("gc::act_backward_sum_kernel", 0.0, s);
  if (dtype == GC_F32) hipLaunchKernelGGL(act_backward_sum_kernel<float>, dim3((unsigned)(p.slabs * chunks)), dim3(kThreads), 0, s, p);
  else hipLaunchKernelGGL(act_backward_sum_kernel<uint16_t>, dim3((unsigned)(p.slabs * chunks)), dim3(kThreads), 0, s, p);
  hipLaunchKernelGGL(channel_sum_final_kernel, dim3((unsigned)((C + 15) / 16)), dim3(kThreads), 0, s, p.part, colsum, p.slabs, C);
  return check_hip(hipGetLastError(), "activation backward launch");
}

int gc_batchnorm_stats(const void* x, float* mean, float* var_unbiased, void* workspace, size_t workspace_bytes, int64_t P, int32_t C,
                       int32_t dtype, void* stream) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("batchnorm: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (P <= 0 || C <= 0 || P >= (1ll << 31)) { set_error("batchnorm: bad shape"); return GC_BAD_ARGUMENT; }
  if (C % 4 != 0) { set_error("batchnorm: channels must be a multiple of 4 (C=%d)", C); return GC_UNSUPPORTED_GEOMETRY; }
  if (x == nullptr || mean == nullptr || var_unbiased == nullptr || workspace == nullptr) { set_error("batchnorm: null pointer"); return GC_BAD_ARGUMENT; }
  if (misaligned(x, dtype == GC_F32 ? 15 : 7) || misaligned(workspace, 15)) { set_error("batchnorm: misaligned pointer"); return GC_BAD_ARGUMENT; }
  const size_t need = bn_workspace(P, C, 0);
  if (workspace_bytes < need) { set_error("batchnorm: workspace of %zu bytes given, %zu needed", workspace_bytes, need); return GC_WORKSPACE_TOO_SMALL; }
  BnParams p{};
  p.x = x; p.save_mean = mean; p.save_var = var_unbiased;
  p.P = (int32_t)P; p.C = C; p.eps = 1.f; p.training = 1;
  bn_geometry(p);
  p.part = (float*)workspace;
  p.coef = p.part + (size_t)p.slabs * C * 2;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (C / 4 + kQB - 1) / kQB;
  {
    ProfScope ps("gc::bn_partial_kernel<0>", 0.0, s);
    if (dtype == GC_F32) hipLaunchKernelGGL((bn_partial_kernel<0, float>), dim3((unsigned)(p.slabs * chunks)), dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL((bn_partial_kernel<0, uint16_t>), dim3((unsigned)(p.slabs * chunks)), dim3(kThreads), 0, s, p);
  }
  { ProfScope ps("gc::bn_finalize_kernel<0>", 0.0, s); hipLaunchKernelGGL(bn_finalize_kernel<0>, dim3((C + kFinCh - 1) / kFinCh), dim3(kThreads), 0, s, p); }
  return check_hip(hipGetLastError(), "batchnorm statistics launch");
}

int gc_batchnorm_act_bwd_sums(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_rstd, float* ggamma, float* gbeta, void* workspace, size_t workspace_bytes, int64_t P,
                              int32_t C, int32_t dtype, int32_t act, float slope, void* stream) {
  return bn_backward(x, gy, gamma, beta, save_mean, save_rstd, nullptr, nullptr, ggamma, gbeta, workspace, workspace_bytes, P, C, dtype, act,
                     slope, stream, 0, 1);
}

int gc_batchnorm_act_bwd_apply(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_rstd, const float* means, void* gx, void* gx_bf16, int64_t P, int32_t C, int32_t dtype,
                               int32_t act, float slope, void* stream) {
  return bn_backward(x, gy, gamma, beta, save_mean, save_rstd, gx, gx_bf16, nullptr, nullptr, nullptr, 0, P, C, dtype, act, slope, stream, 0,
                     2, means);
}

int gc_batchnorm_update_running(float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* mean,
                                const float* var_unbiased, int32_t C, float momentum, void* stream) {
  if (C <= 0 || running_mean == nullptr || running_var == nullptr || mean == nullptr || var_unbiased == nullptr) {
    set_error("batchnorm: bad argument to the running-statistics update"); return GC_BAD_ARGUMENT;
  }
  ProfScope ps("gc::bn_update_running_kernel", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_update_running_kernel, dim3((C + kThreads - 1) / kThreads), dim3(kThreads), 0, (hipStream_t)stream, running_mean,
                     running_var, num_batches_tracked, mean, var_unbiased, C, momentum);
  return check_hip(hipGetLastError(), "batchnorm running-statistics launch");
}

}  // extern "C"
