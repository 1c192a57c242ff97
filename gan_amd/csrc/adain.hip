// StyleGAN's per-layer chain behind every synthesis convolution (reference StyleGAN/model.py:179-208,249-261) as one
// kernel per direction, NHWC fp32 or bf16 tensors, fp32 arithmetic:
//
//   v = x + sf[c] * noise[n, hw]            (NoiseInput: one noise image per sample, one learned scale per channel)
//   t = lrelu(v)                            (LeakyReLU 0.2)
//   th = (t - mu) * r                       (InstanceNorm2d: per (n, c) mean / rsqrt(var + eps) over H*W, biased)
//   out = ys[n, c] * th + yb[n, c]          (AdaIN style from the affine layer)
//
//   backward (first order; the generator is never differentiated twice):
//     g_yb[n,c] = sum_hw g,  g_ys[n,c] = sum_hw g * th,
//     g_v = lrelu'(v) * r * ys * (g - mean(g) - th * mean(g * th)),  g_x = g_v,  g_sf[c] = sum_{n,hw} g_v * noise
//   (g_sf is returned per sample, [N, C]; the caller sums over samples.  The noise image gets no gradient.)
//
// Small images: one workgroup owns one sample and a run of up to 64 channels (16-byte accesses along the channel axis x
// lanes over pixels), as the InstanceNorm kernels do; the slab is re-read from L2.  Forward: three sweeps (sum, centred
// squares, write); it returns mu and r, so the backward pass needs two (sums, write).
// Large images (a sample's plane split into `sps` slabs of pixels so that the launch fills the chip): a partial launch
// writes per-slab (mean, M2) or (sum g, sum g*th); the apply launch merges the slabs of its sample in a fixed order
// (Chan's update in the forward pass) and sweeps its own slab; the backward pass ends with a small launch that adds the
// per-slab noise-scale sums.  Workspace: gc_noise_lrelu_adain_workspace_bytes.
#include "common.h"

namespace gc {
namespace {

struct AdainParams {
  const void* x;
  const void* g;        // backward: gradient of out
  void* out;            // out | g_x
  const float* noise;   // [N, HW]
  const float* sf;      // [C]
  const float* ys;      // [N, C]
  const float* yb;      // [N, C]
  float* mu;            // [N, C]  written by the forward pass, read by the backward pass
  float* r;             // [N, C]
  float* g_ys;          // [N, C]
  float* g_yb;          // [N, C]
  float* g_sf;          // [N, C]
  int32_t N, HW, C, qb_log2;
  float eps, slope;
  float* part;          // slab mode: [N][sps][C][2] partial statistics, then [N][sps][C] noise-scale sums
  int32_t sps, slab_pix;
};

__device__ __forceinline__ f32x4 wg_sum(f32x4 v, f32x4* red, int tq, int tl, int QB, int PL) {
  red[tl * QB + tq] = v;
  __syncthreads();
  if (tl == 0) {
    f32x4 s = v;
    for (int j = 1; j < PL; j++) s += red[j * QB + tq];
    red[tq] = s;
  }
  __syncthreads();
  const f32x4 r = red[tq];
  __syncthreads();
  return r;
}

template <bool BWD, typename T>
__global__ __launch_bounds__(kThreads) void adain_kernel(const AdainParams p) {
  __shared__ f32x4 red[kThreads];
  const int QB = 1 << p.qb_log2, PL = kThreads >> p.qb_log2;
  const int t = threadIdx.x, tq = t & (QB - 1), tl = t >> p.qb_log2;
  const int chunks = (p.C / 4 + QB - 1) / QB;
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = (chunk * QB + tq) * 4;
  const bool cok = c < p.C;
  const size_t base = (size_t)n * p.HW * p.C + (cok ? c : 0);
  const size_t nc = (size_t)n * p.C + (cok ? c : 0);
  const float* nz = p.noise + (size_t)n * p.HW;
  const float inv_n = 1.f / (float)p.HW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sf = cok ? *reinterpret_cast<const f32x4*>(p.sf + c) : zero;
  auto pre = [&](int pix) -> f32x4 { return load4<T>(p.x, base + (size_t)pix * p.C) + sf * nz[pix]; };
  auto lrelu = [&](f32x4 v) -> f32x4 {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
    return o;
  };

  if (!BWD) {
    f32x4 s = zero;
    if (cok)
      for (int pix = tl; pix < p.HW; pix += PL) s += lrelu(pre(pix));
    const f32x4 mu = wg_sum(s, red, tq, tl, QB, PL) * inv_n;
    f32x4 s2 = zero;
    if (cok)
      for (int pix = tl; pix < p.HW; pix += PL) {
        const f32x4 d = lrelu(pre(pix)) - mu;
        s2 += d * d;
      }
    s2 = wg_sum(s2, red, tq, tl, QB, PL);
    if (!cok) return;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = rsqrtf(s2[e] * inv_n + p.eps);
    if (tl == 0) {
      *reinterpret_cast<f32x4*>(p.mu + nc) = mu;
      *reinterpret_cast<f32x4*>(p.r + nc) = r;
    }
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.ys + nc) * r, b = *reinterpret_cast<const f32x4*>(p.yb + nc);
    for (int pix = tl; pix < p.HW; pix += PL) store4<T>(p.out, base + (size_t)pix * p.C, a * (lrelu(pre(pix)) - mu) + b);
  } else {
    const f32x4 mu = cok ? *reinterpret_cast<const f32x4*>(p.mu + nc) : zero, r = cok ? *reinterpret_cast<const f32x4*>(p.r + nc) : zero;
    f32x4 sg = zero, sgt = zero;
    if (cok)
      for (int pix = tl; pix < p.HW; pix += PL) {
        const f32x4 g = load4<T>(p.g, base + (size_t)pix * p.C);
        sg += g;
        sgt += g * ((lrelu(pre(pix)) - mu) * r);
      }
    sg = wg_sum(sg, red, tq, tl, QB, PL);
    sgt = wg_sum(sgt, red, tq, tl, QB, PL);
    f32x4 snz = zero;
    if (cok) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p.ys + nc) * r, gbar = sg * inv_n, tbar = sgt * inv_n;
      for (int pix = tl; pix < p.HW; pix += PL) {
        const f32x4 v = pre(pix);
        const f32x4 th = (lrelu(v) - mu) * r;
        f32x4 gv = a * (load4<T>(p.g, base + (size_t)pix * p.C) - gbar - th * tbar);
#pragma unroll
        for (int e = 0; e < 4; e++) gv[e] = v[e] > 0.f ? gv[e] : gv[e] * p.slope;
        store4<T>(p.out, base + (size_t)pix * p.C, gv);
        snz += gv * nz[pix];
      }
    }
    snz = wg_sum(snz, red, tq, tl, QB, PL);
    if (cok && tl == 0) {
      *reinterpret_cast<f32x4*>(p.g_yb + nc) = sg;
      *reinterpret_cast<f32x4*>(p.g_ys + nc) = sgt;
      *reinterpret_cast<f32x4*>(p.g_sf + nc) = snz;
    }
  }
}

// ---- slab mode ------------------------------------------------------------------------------------------------
// PHASE 0: partial statistics of one slab;  PHASE 1: merge the sample's slabs, then apply on this slab
template <bool BWD, int PHASE, typename T>
__global__ __launch_bounds__(kThreads) void adain_slab_kernel(const AdainParams p) {
  __shared__ f32x4 red[kThreads];
  const int QB = 1 << p.qb_log2, PL = kThreads >> p.qb_log2;
  const int t = threadIdx.x, tq = t & (QB - 1), tl = t >> p.qb_log2;
  const int chunks = (p.C / 4 + QB - 1) / QB;
  const int chunk = blockIdx.x % chunks, ns = blockIdx.x / chunks;      // ns = n * sps + slab
  const int n = ns / p.sps, slab = ns - n * p.sps;
  const int c = (chunk * QB + tq) * 4;
  const bool cok = c < p.C;
  const int p0 = slab * p.slab_pix, p1 = min(p.HW, p0 + p.slab_pix);
  const size_t base = (size_t)n * p.HW * p.C + (cok ? c : 0);
  const size_t nc = (size_t)n * p.C + (cok ? c : 0);
  const float* nz = p.noise + (size_t)n * p.HW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 sf = cok ? *reinterpret_cast<const f32x4*>(p.sf + c) : zero;
  auto pre = [&](int pix) -> f32x4 { return load4<T>(p.x, base + (size_t)pix * p.C) + sf * nz[pix]; };
  auto lrelu = [&](f32x4 v) -> f32x4 {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = v[e] > 0.f ? v[e] : v[e] * p.slope;
    return o;
  };
  float* part = p.part + ((size_t)ns * p.C + (cok ? c : 0)) * 2;
  const float* parts_n = p.part + ((size_t)n * p.sps * p.C + (cok ? c : 0)) * 2;   // slab s of this sample: + s * C * 2

  if (PHASE == 0) {
    f32x4 a = zero, b = zero;
    if (!BWD) {
      if (cok)
        for (int pix = p0 + tl; pix < p1; pix += PL) a += lrelu(pre(pix));
      a = wg_sum(a, red, tq, tl, QB, PL) * (1.f / (float)(p1 - p0));      // slab mean
      if (cok)
        for (int pix = p0 + tl; pix < p1; pix += PL) {
          const f32x4 d = lrelu(pre(pix)) - a;
          b += d * d;
        }
      b = wg_sum(b, red, tq, tl, QB, PL);                                   // slab M2
    } else {
      const f32x4 mu = cok ? *reinterpret_cast<const f32x4*>(p.mu + nc) : zero, r = cok ? *reinterpret_cast<const f32x4*>(p.r + nc) : zero;
      if (cok)
        for (int pix = p0 + tl; pix < p1; pix += PL) {
          const f32x4 g = load4<T>(p.g, base + (size_t)pix * p.C);
          a += g;
          b += g * ((lrelu(pre(pix)) - mu) * r);
        }
      a = wg_sum(a, red, tq, tl, QB, PL);
      b = wg_sum(b, red, tq, tl, QB, PL);
    }
    if (cok && tl == 0) {
      *reinterpret_cast<f32x4*>(part) = f32x4{a[0], b[0], a[1], b[1]};
      *reinterpret_cast<f32x4*>(part + 4) = f32x4{a[2], b[2], a[3], b[3]};
    }
    return;
  }

  if (!cok) {           // (keeps the barriers of wg_sum below matched: BWD only)
    if (BWD) wg_sum(zero, red, tq, tl, QB, PL);
    return;
  }
  f32x4 a = zero, b = zero;      // forward: mean, M2;  backward: sum g, sum g*th
  float cnt = 0.f;
  for (int s2 = 0; s2 < p.sps; s2++) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(parts_n + (size_t)s2 * p.C * 2), w = *reinterpret_cast<const f32x4*>(parts_n + (size_t)s2 * p.C * 2 + 4);
    const f32x4 va = {u[0], u[2], w[0], w[2]}, vb = {u[1], u[3], w[1], w[3]};
    if (!BWD) {
      const float nb = (float)(min(p.HW, (s2 + 1) * p.slab_pix) - s2 * p.slab_pix), tot = cnt + nb;
      const f32x4 delta = va - a;
      a += delta * (nb / tot);
      b += vb + delta * delta * (cnt * nb / tot);
      cnt = tot;
    } else {
      a += va;
      b += vb;
    }
  }
  const float inv_n = 1.f / (float)p.HW;
  if (!BWD) {
    const f32x4 mu = a;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; e++) r[e] = rsqrtf(b[e] * inv_n + p.eps);
    if (slab == 0 && tl == 0) {
      *reinterpret_cast<f32x4*>(p.mu + nc) = mu;
      *reinterpret_cast<f32x4*>(p.r + nc) = r;
    }
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.ys + nc) * r, be = *reinterpret_cast<const f32x4*>(p.yb + nc);
    for (int pix = p0 + tl; pix < p1; pix += PL) store4<T>(p.out, base + (size_t)pix * p.C, ga * (lrelu(pre(pix)) - mu) + be);
  } else {
    const f32x4 mu = *reinterpret_cast<const f32x4*>(p.mu + nc), r = *reinterpret_cast<const f32x4*>(p.r + nc);
    const f32x4 ga = *reinterpret_cast<const f32x4*>(p.ys + nc) * r, gbar = a * inv_n, tbar = b * inv_n;
    f32x4 snz = zero;
    for (int pix = p0 + tl; pix < p1; pix += PL) {
      const f32x4 v = pre(pix);
      const f32x4 th = (lrelu(v) - mu) * r;
      f32x4 gv = ga * (load4<T>(p.g, base + (size_t)pix * p.C) - gbar - th * tbar);
#pragma unroll
      for (int e = 0; e < 4; e++) gv[e] = v[e] > 0.f ? gv[e] : gv[e] * p.slope;
      store4<T>(p.out, base + (size_t)pix * p.C, gv);
      snz += gv * nz[pix];
    }
    snz = wg_sum(snz, red, tq, tl, QB, PL);
    if (tl == 0) {
      *reinterpret_cast<f32x4*>(p.part + (size_t)p.N * p.sps * p.C * 2 + (size_t)ns * p.C + c) = snz;
      if (slab == 0) {
        *reinterpret_cast<f32x4*>(p.g_yb + nc) = a;
        *reinterpret_cast<f32x4*>(p.g_ys + nc) = b;
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void adain_noise_sum_kernel(const AdainParams p) {
  const int i = blockIdx.x * kThreads + threadIdx.x;      // (n, c)
  if (i >= p.N * p.C) return;
  const int n = i / p.C, c = i - n * p.C;
  const float* src = p.part + (size_t)p.N * p.sps * p.C * 2 + (size_t)n * p.sps * p.C + c;
  float s = 0.f;
  for (int k = 0; k < p.sps; k++) s += src[(size_t)k * p.C];
  p.g_sf[i] = s;
}

// Slabs per sample: about 2048 workgroups in all, at least 256 pixels per slab, at most 64 slabs.
void slab_geometry(AdainParams& p) {
  const int QB = p.C / 4 < 16 ? p.C / 4 : 16;
  const long chunks = (p.C / 4 + QB - 1) / QB;
  long sps = 2048 / ((long)p.N * chunks);
  if (sps > p.HW / 256) sps = p.HW / 256;
  if (sps > 64) sps = 64;
  if (sps < 1) sps = 1;
  p.slab_pix = (int)((p.HW + sps - 1) / sps);
  p.sps = (p.HW + p.slab_pix - 1) / p.slab_pix;
}

size_t workspace_bytes(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  AdainParams p{};
  p.N = N; p.HW = HW; p.C = C;
  slab_geometry(p);
  return p.sps > 1 ? (size_t)N * p.sps * C * 3 * sizeof(float) : 0;
}

int launch(bool bwd, AdainParams& p, int dtype, hipStream_t stream, void* workspace, size_t workspace_bytes_given) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("adain: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (p.N < 0 || p.HW <= 0 || p.C <= 0) { set_error("adain: bad shape"); return GC_BAD_ARGUMENT; }
  if (p.C % 4 != 0) { set_error("adain: channels must be a multiple of 4 (C=%d)", p.C); return GC_UNSUPPORTED_GEOMETRY; }
  if (!(p.slope > 0.f)) { set_error("adain: LeakyReLU slope must be positive"); return GC_UNSUPPORTED_GEOMETRY; }
  if (p.N == 0) return GC_OK;
  const void* ptrs[] = {p.x, p.out, p.noise, p.sf, p.ys, p.yb, p.mu, p.r, bwd ? p.g : p.x, bwd ? p.g_ys : p.mu, bwd ? p.g_yb : p.mu, bwd ? p.g_sf : p.mu};
  for (auto q : ptrs)
    if (q == nullptr || (reinterpret_cast<uintptr_t>(q) & 7u)) { set_error("adain: null or misaligned pointer"); return GC_BAD_ARGUMENT; }
  if ((reinterpret_cast<uintptr_t>(p.sf) | reinterpret_cast<uintptr_t>(p.ys) | reinterpret_cast<uintptr_t>(p.yb) | reinterpret_cast<uintptr_t>(p.mu) |
       reinterpret_cast<uintptr_t>(p.r)) & 15u) { set_error("adain: per-channel vectors must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  if (dtype == GC_F32 && ((reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(bwd ? p.g : p.x)) & 15u)) {
    set_error("adain: fp32 tensors must be 16-byte aligned"); return GC_BAD_ARGUMENT;
  }
  slab_geometry(p);
  if (p.sps > 1) {
    const size_t need = workspace_bytes(p.N, p.HW, p.C);
    if (workspace == nullptr || (reinterpret_cast<uintptr_t>(workspace) & 15u) || workspace_bytes_given < need) {
      set_error("adain: workspace of %zu bytes given, %zu needed (16-byte aligned)", workspace_bytes_given, need);
      return GC_WORKSPACE_TOO_SMALL;
    }
    p.part = (float*)workspace;
    p.qb_log2 = 0;
    while ((1 << p.qb_log2) < p.C / 4 && p.qb_log2 < 4) p.qb_log2++;
    const int QB = 1 << p.qb_log2;
    const dim3 grid((unsigned)p.N * p.sps * ((p.C / 4 + QB - 1) / QB)), block(kThreads);
    ProfScope ps(bwd ? "gc::adain_slab_kernel<bwd>" : "gc::adain_slab_kernel<fwd>", 0.0, stream);
    if (dtype == GC_F32) {
      if (bwd) {
        hipLaunchKernelGGL((adain_slab_kernel<true, 0, float>), grid, block, 0, stream, p);
        hipLaunchKernelGGL((adain_slab_kernel<true, 1, float>), grid, block, 0, stream, p);
      } else {
        hipLaunchKernelGGL((adain_slab_kernel<false, 0, float>), grid, block, 0, stream, p);
        hipLaunchKernelGGL((adain_slab_kernel<false, 1, float>), grid, block, 0, stream, p);
      }
    } else {
      if (bwd) {
        hipLaunchKernelGGL((adain_slab_kernel<true, 0, uint16_t>), grid, block, 0, stream, p);
        hipLaunchKernelGGL((adain_slab_kernel<true, 1, uint16_t>), grid, block, 0, stream, p);
      } else {
        hipLaunchKernelGGL((adain_slab_kernel<false, 0, uint16_t>), grid, block, 0, stream, p);
        hipLaunchKernelGGL((adain_slab_kernel<false, 1, uint16_t>), grid, block, 0, stream, p);
      }
    }
    if (bwd) hipLaunchKernelGGL(adain_noise_sum_kernel, dim3((unsigned)((p.N * p.C + kThreads - 1) / kThreads)), block, 0, stream, p);
    return check_hip(hipGetLastError(), "adain launch");
  }
  p.qb_log2 = 0;
  while ((1 << p.qb_log2) < p.C / 4 && p.qb_log2 < 4) p.qb_log2++;
  while (p.qb_log2 > 2 && (long)p.N * ((p.C / 4 + (1 << p.qb_log2) - 1) >> p.qb_log2) < 512 && p.HW >= 1024) p.qb_log2--;
  const int QB = 1 << p.qb_log2;
  const dim3 grid((unsigned)p.N * ((p.C / 4 + QB - 1) / QB)), block(kThreads);
  ProfScope ps(bwd ? "gc::adain_kernel<bwd>" : "gc::adain_kernel<fwd>", 0.0, stream);
  if (dtype == GC_F32) {
    if (bwd) hipLaunchKernelGGL((adain_kernel<true, float>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((adain_kernel<false, float>), grid, block, 0, stream, p);
  } else {
    if (bwd) hipLaunchKernelGGL((adain_kernel<true, uint16_t>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((adain_kernel<false, uint16_t>), grid, block, 0, stream, p);
  }
  return check_hip(hipGetLastError(), "adain launch");
}

}  // namespace
}  // namespace gc

extern "C" size_t gc_noise_lrelu_adain_workspace_bytes(int32_t N, int32_t HW, int32_t C) { return gc::workspace_bytes(N, HW, C); }

extern "C" int gc_noise_lrelu_adain_fwd(const void* x, const float* noise, const float* sf, const float* ys, const float* yb, void* out,
                                        float* mu, float* r, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW, int32_t C,
                                        int32_t dtype, float eps, float slope, void* stream) {
  gc::AdainParams p{};
  p.x = x; p.noise = noise; p.sf = sf; p.ys = ys; p.yb = yb; p.out = out; p.mu = mu; p.r = r;
  p.N = N; p.HW = HW; p.C = C; p.eps = eps; p.slope = slope;
  return gc::launch(false, p, dtype, (hipStream_t)stream, workspace, workspace_bytes);
}

extern "C" int gc_noise_lrelu_adain_bwd(const void* x, const float* noise, const float* sf, const float* ys, const float* yb,
                                        const float* mu, const float* r, const void* gout, void* gx, float* g_sf, float* g_ys, float* g_yb,
                                        void* workspace, size_t workspace_bytes, int32_t N, int32_t HW, int32_t C, int32_t dtype,
                                        float slope, void* stream) {
  gc::AdainParams p{};
  p.x = x; p.noise = noise; p.sf = sf; p.ys = ys; p.yb = yb; p.mu = const_cast<float*>(mu); p.r = const_cast<float*>(r);
  p.g = gout; p.out = gx; p.g_sf = g_sf; p.g_ys = g_ys; p.g_yb = g_yb;
  p.N = N; p.HW = HW; p.C = C; p.eps = 0.f; p.slope = slope;
  return gc::launch(true, p, dtype, (hipStream_t)stream, workspace, workspace_bytes);
}
