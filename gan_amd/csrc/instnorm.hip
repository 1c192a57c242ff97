// InstanceNorm2d (affine = false, biased variance) fused with LeakyReLU, NHWC fp32 or bf16 (fp32 arithmetic): forward, backward and the
// backward of the backward (needed by the WGAN-GP gradient penalty, which differentiates the critic twice).
//
//   forward       z  = lrelu(xh),  xh = (x - mu) * r,  mu / r = per-(n,c) mean and rsqrt(var + eps) over H*W
//   backward      gx = r * P(gy),  gy = gz * m,  m = (xh > 0 ? 1 : slope),  P(v) = v - mean(v) - xh * mean(v * xh)
//   2nd backward  given u (the gradient arriving at gx):
//                   g_gz = m * r * P(u)
//                   g_x  = -(r^2 / n) * [ (a - b*c/n) * xh + c * P(u) + b * P(gy) ]
//                 with n = H*W, a = sum(u*gy) - n*mean(u)*mean(gy), b = sum(u*xh), c = sum(gy*xh)
//                 (P is self-adjoint and LeakyReLU is piecewise linear, so no further terms appear).
//
// One workgroup owns one sample and a run of 64 channels: 16 lanes across channel quads (16-byte accesses,
// coalesced along the NHWC channel axis) x 16 lanes across pixels.  Statistics are recomputed from x in every
// kernel (mean first, then centred sums: no E[x^2]-E[x]^2 cancellation); the per-workgroup slab (H*W x 64
// channels) is re-read from L2.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace gc {

struct InstNormParams {   // tensors are fp32 or bf16 (template parameter of the kernel), all of one type
  const void* x;
  const void* gz;
  const void* u;
  void* out0;    // z | gx | g_x
  void* out1;    // - | -  | g_gz
  uint16_t* out0_bf16;   // optional bf16 copies of out0 / out1
  uint16_t* out1_bf16;
  int32_t N, HW, C;
  int32_t qb_log2;   // log2 of channel quads per workgroup (<= 4)
  float eps, slope;
  int32_t act;       // GC_ACT_NONE or GC_ACT_LRELU
};

__device__ __forceinline__ f32x4 block_sum(f32x4 v, f32x4* red, int tq, int tl, int QB, int PL) {
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

// CACHED: a thread's share of the slab (at most kCachePix pixels x 4 channels per tensor) stays in registers, in storage
// form, from the first sweep on: every tensor is read once instead of two or three times.  Same arithmetic in the same
// order as the re-reading form (bit-identical results).
constexpr int kCachePix = 16;

template <typename T>
struct RawQuad {          // 4 channels in storage form
  typedef typename std::conditional<sizeof(T) == 2, u32x2, f32x4>::type type;
  __device__ __forceinline__ static type load(const void* base, size_t idx) {
    if constexpr (sizeof(T) == 2) return *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(base) + idx);
    else return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  }
  __device__ __forceinline__ static f32x4 widen(type v) {
    if constexpr (sizeof(T) == 2) return widen_bf16x4(v);
    else return v;
  }
};

template <int MODE, typename T, bool CACHED>
__global__ __launch_bounds__(kThreads) void instnorm_kernel(const InstNormParams p) {
  __shared__ f32x4 red[kThreads];
  typedef RawQuad<T> RQ;
  constexpr int KC = CACHED ? kCachePix : 1;
  typename RQ::type xr[KC], gr[MODE >= 1 ? KC : 1], ur[MODE == 2 ? KC : 1];
  const int QB = 1 << p.qb_log2, PL = kThreads >> p.qb_log2;
  const int t = threadIdx.x, tq = t & (QB - 1), tl = t >> p.qb_log2;
  const int chunks = (p.C / 4 + QB - 1) / QB;
  const int n = blockIdx.x / chunks, chunk = blockIdx.x % chunks;
  const int c = (chunk * QB + tq) * 4;
  const bool cok = c < p.C;
  const size_t base = (size_t)n * p.HW * p.C + (cok ? c : 0);
  const float inv_n = 1.f / (float)p.HW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  auto ld = [&](const void* ptr, int pix) -> f32x4 { return load4<T>(ptr, base + (size_t)pix * p.C); };
  if (CACHED && cok) {      // all loads of the workgroup's slab are issued here, before the first reduction
#pragma unroll
    for (int i = 0; i < KC; i++) {
      const int pix = tl + i * PL;
      if (pix < p.HW) {
        xr[i] = RQ::load(p.x, base + (size_t)pix * p.C);
        if (MODE >= 1) gr[i] = RQ::load(p.gz, base + (size_t)pix * p.C);
        if (MODE == 2) ur[i] = RQ::load(p.u, base + (size_t)pix * p.C);
      }
    }
  }
  // for_pixels(f): f(pix, x, gz, u) over this thread's pixels in ascending order, from the cache or from memory
  auto for_pixels = [&](auto&& f) {
    if (CACHED) {
#pragma unroll
      for (int i = 0; i < KC; i++) {
        const int pix = tl + i * PL;
        if (pix < p.HW) f(pix, RQ::widen(xr[i]), MODE >= 1 ? RQ::widen(gr[MODE >= 1 ? i : 0]) : f32x4{0.f, 0.f, 0.f, 0.f},
                          MODE == 2 ? RQ::widen(ur[MODE == 2 ? i : 0]) : f32x4{0.f, 0.f, 0.f, 0.f});
      }
    } else {
      for (int pix = tl; pix < p.HW; pix += PL)
        f(pix, ld(p.x, pix), MODE >= 1 ? ld(p.gz, pix) : f32x4{0.f, 0.f, 0.f, 0.f}, MODE == 2 ? ld(p.u, pix) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  };
  auto mask = [&](f32x4 d) -> f32x4 {
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; e++) m[e] = (p.act == GC_ACT_LRELU && !(d[e] > 0.f)) ? p.slope : 1.f;
    return m;
  };

  f32x4 s = zero;
  if (cok) for_pixels([&](int, f32x4 xv, f32x4, f32x4) { s += xv; });
  const f32x4 mu = block_sum(s, red, tq, tl, QB, PL) * inv_n;

  f32x4 s2 = zero, sg = zero, sgd = zero, su = zero, sud = zero, sug = zero;
  if (cok)
    for_pixels([&](int, f32x4 xv, f32x4 gv, f32x4 uu) {
      const f32x4 d = xv - mu;
      s2 += d * d;
      if (MODE >= 1) {
        const f32x4 gy = gv * mask(d);
        sg += gy;
        sgd += gy * d;
        if (MODE == 2) {
          su += uu;
          sud += uu * d;
          sug += uu * gy;
        }
      }
    });
  s2 = block_sum(s2, red, tq, tl, QB, PL);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; e++) r[e] = rsqrtf(s2[e] * inv_n + p.eps);
  if (MODE >= 1) {
    sg = block_sum(sg, red, tq, tl, QB, PL);
    sgd = block_sum(sgd, red, tq, tl, QB, PL);
  }
  if (MODE == 2) {
    su = block_sum(su, red, tq, tl, QB, PL);
    sud = block_sum(sud, red, tq, tl, QB, PL);
    sug = block_sum(sug, red, tq, tl, QB, PL);
  }
  if (!cok) return;

  if (MODE == 0) {
    for_pixels([&](int pix, f32x4 xv, f32x4, f32x4) {
      const f32x4 d = xv - mu;
      const f32x4 z = d * r * mask(d);
      store4<T>(p.out0, base + (size_t)pix * p.C, z);
      if (p.out0_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out0_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(z);
    });
  } else if (MODE == 1) {
    const f32x4 gbar = sg * inv_n, cbar = sgd * r * inv_n;   // mean(gy), mean(gy * xh)
    for_pixels([&](int pix, f32x4 xv, f32x4 gv, f32x4) {
      const f32x4 d = xv - mu;
      const f32x4 gy = gv * mask(d);
      const f32x4 gx = r * (gy - gbar - d * r * cbar);
      store4<T>(p.out0, base + (size_t)pix * p.C, gx);
      if (p.out0_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out0_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(gx);
    });
  } else {
    const float nn = (float)p.HW;
    const f32x4 gbar = sg * inv_n, ubar = su * inv_n;
    const f32x4 c_ = sgd * r, b_ = sud * r;                  // sum(gy * xh), sum(u * xh)
    const f32x4 a_ = sug - nn * ubar * gbar;
    const f32x4 k0 = a_ - b_ * c_ * inv_n;
    const f32x4 scale = -(r * r) * inv_n;
    for_pixels([&](int pix, f32x4 xv, f32x4 gv, f32x4 uu) {
      const f32x4 d = xv - mu;
      const f32x4 m = mask(d);
      const f32x4 xh = d * r;
      const f32x4 gy = gv * m;
      const f32x4 pu = uu - ubar - xh * (b_ * inv_n);
      const f32x4 pg = gy - gbar - xh * (c_ * inv_n);
      const f32x4 o0 = scale * (k0 * xh + c_ * pu + b_ * pg), o1 = m * r * pu;
      store4<T>(p.out0, base + (size_t)pix * p.C, o0);
      store4<T>(p.out1, base + (size_t)pix * p.C, o1);
      if (p.out0_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out0_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(o0);
      if (p.out1_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out1_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(o1);
    });
  }
}

// Images of up to 256 pixels (the critics' 16 x 16, 8 x 8 and 4 x 4 maps): one WAVE owns a sample and a run of channels --
// QL lanes across channel quads x PL lanes across pixels (16 x 4 up to 64 pixels, 4 x 16 up to 256), every lane keeping
// its pixels (at most 16) in registers -- so the sums over the pixels are lane exchanges and the kernel has no LDS traffic
// and no barrier.  The workgroup form above spends a slab of 2-8 KB on five to seven workgroup-wide reductions (three
// barriers each): 0.8-1.1 TB/s on [512, 512, 4, 4].  A workgroup is four independent waves on neighbouring channel runs
// (with 4 quads per wave they share every 128-byte line they touch).
template <int PL>
__device__ __forceinline__ f32x4 pixel_lanes_sum(f32x4 v) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
#pragma unroll
    for (int o = 64 / PL; o < 64; o <<= 1) v[e] += __shfl_xor(v[e], o, 64);
  }
  return v;
}

template <int MODE, typename T, int KC, int PL>      // KC: pixels per lane, PL: lanes across pixels (KC * PL pixels at most)
__global__ __launch_bounds__(kThreads) void instnorm_wave_kernel(const InstNormParams p) {
  typedef RawQuad<T> RQ;
  constexpr int QL = 64 / PL;
  typename RQ::type xr[KC], gr[MODE >= 1 ? KC : 1], ur[MODE == 2 ? KC : 1];
  const int lane = threadIdx.x & 63, tq = lane & (QL - 1), tl = lane / QL;
  const int chunks = (p.C / 4 + QL - 1) / QL;
  const long slab = (long)blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6);
  if (slab >= (long)p.N * chunks) return;          // (whole waves leave: the exchanges below stay inside a wave)
  const int n = (int)(slab / chunks), chunk = (int)(slab % chunks);
  const int c = (chunk * QL + tq) * 4;
  const bool cok = c < p.C;
  const size_t base = (size_t)n * p.HW * p.C + (cok ? c : 0);
  const float inv_n = 1.f / (float)p.HW;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < KC; i++) {
    const int pix = tl + i * PL;
    if (cok && pix < p.HW) {
      xr[i] = RQ::load(p.x, base + (size_t)pix * p.C);
      if (MODE >= 1) gr[i] = RQ::load(p.gz, base + (size_t)pix * p.C);
      if (MODE == 2) ur[i] = RQ::load(p.u, base + (size_t)pix * p.C);
    }
  }
  auto for_pixels = [&](auto&& f) {
#pragma unroll
    for (int i = 0; i < KC; i++) {
      const int pix = tl + i * PL;
      if (cok && pix < p.HW) f(pix, RQ::widen(xr[i]), MODE >= 1 ? RQ::widen(gr[MODE >= 1 ? i : 0]) : f32x4{0.f, 0.f, 0.f, 0.f},
                               MODE == 2 ? RQ::widen(ur[MODE == 2 ? i : 0]) : f32x4{0.f, 0.f, 0.f, 0.f});
    }
  };
  auto mask = [&](f32x4 d) -> f32x4 {
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; e++) m[e] = (p.act == GC_ACT_LRELU && !(d[e] > 0.f)) ? p.slope : 1.f;
    return m;
  };

  f32x4 s = zero;
  for_pixels([&](int, f32x4 xv, f32x4, f32x4) { s += xv; });
  const f32x4 mu = pixel_lanes_sum<PL>(s) * inv_n;

  f32x4 s2 = zero, sg = zero, sgd = zero, su = zero, sud = zero, sug = zero;
  for_pixels([&](int, f32x4 xv, f32x4 gv, f32x4 uu) {
    const f32x4 d = xv - mu;
    s2 += d * d;
    if (MODE >= 1) {
      const f32x4 gy = gv * mask(d);
      sg += gy;
      sgd += gy * d;
      if (MODE == 2) {
        su += uu;
        sud += uu * d;
        sug += uu * gy;
      }
    }
  });
  s2 = pixel_lanes_sum<PL>(s2);
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; e++) r[e] = rsqrtf(s2[e] * inv_n + p.eps);
  if (MODE >= 1) { sg = pixel_lanes_sum<PL>(sg); sgd = pixel_lanes_sum<PL>(sgd); }
  if (MODE == 2) { su = pixel_lanes_sum<PL>(su); sud = pixel_lanes_sum<PL>(sud); sug = pixel_lanes_sum<PL>(sug); }

  if (MODE == 0) {
    for_pixels([&](int pix, f32x4 xv, f32x4, f32x4) {
      const f32x4 d = xv - mu;
      const f32x4 z = d * r * mask(d);
      store4<T>(p.out0, base + (size_t)pix * p.C, z);
      if (p.out0_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out0_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(z);
    });
  } else if (MODE == 1) {
    const f32x4 gbar = sg * inv_n, cbar = sgd * r * inv_n;
    for_pixels([&](int pix, f32x4 xv, f32x4 gv, f32x4) {
      const f32x4 d = xv - mu;
      const f32x4 gy = gv * mask(d);
      const f32x4 gx = r * (gy - gbar - d * r * cbar);
      store4<T>(p.out0, base + (size_t)pix * p.C, gx);
      if (p.out0_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out0_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(gx);
    });
  } else {
    const float nn = (float)p.HW;
    const f32x4 gbar = sg * inv_n, ubar = su * inv_n;
    const f32x4 c_ = sgd * r, b_ = sud * r;
    const f32x4 a_ = sug - nn * ubar * gbar;
    const f32x4 k0 = a_ - b_ * c_ * inv_n;
    const f32x4 scale = -(r * r) * inv_n;
    for_pixels([&](int pix, f32x4 xv, f32x4 gv, f32x4 uu) {
      const f32x4 d = xv - mu;
      const f32x4 m = mask(d);
      const f32x4 xh = d * r;
      const f32x4 gy = gv * m;
      const f32x4 pu = uu - ubar - xh * (b_ * inv_n);
      const f32x4 pg = gy - gbar - xh * (c_ * inv_n);
      const f32x4 o0 = scale * (k0 * xh + c_ * pu + b_ * pg), o1 = m * r * pu;
      store4<T>(p.out0, base + (size_t)pix * p.C, o0);
      store4<T>(p.out1, base + (size_t)pix * p.C, o1);
      if (p.out0_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out0_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(o0);
      if (p.out1_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out1_bf16 + base + (size_t)pix * p.C) = cvt_bf16x4(o1);
    });
  }
}

static int launch_instnorm(int mode, int dtype, const void* x, const void* gz, const void* u, void* out0, void* out1, void* out0_bf16,
                           void* out1_bf16, int N, int HW, int C, float eps, float slope, int act, hipStream_t stream) {
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("instnorm: unknown dtype"); return GC_BAD_ARGUMENT; }
  if (dtype == GC_BF16 && (out0_bf16 != nullptr || out1_bf16 != nullptr)) { set_error("instnorm: bf16 tensors need no bf16 copy"); return GC_BAD_ARGUMENT; }
  if (N < 0 || HW <= 0 || C <= 0) { set_error("instnorm: bad shape"); return GC_BAD_ARGUMENT; }
  if (act != GC_ACT_NONE && act != GC_ACT_LRELU) { set_error("instnorm: activation must be none or LeakyReLU"); return GC_BAD_ARGUMENT; }
  if (act == GC_ACT_LRELU && !(slope > 0.f)) { set_error("instnorm: LeakyReLU slope must be positive"); return GC_UNSUPPORTED_GEOMETRY; }
  if (C % 4 != 0) { set_error("instnorm: channels must be a multiple of 4 (C=%d)", C); return GC_UNSUPPORTED_GEOMETRY; }
  if (N == 0) return GC_OK;
  const void* ptrs[5] = {x, mode >= 1 ? gz : x, mode == 2 ? u : x, out0, mode == 2 ? out1 : out0};
  for (auto q : ptrs)
    if (q == nullptr || (reinterpret_cast<uintptr_t>(q) & (dtype == GC_F32 ? 15u : 7u))) { set_error("instnorm: null or misaligned pointer"); return GC_BAD_ARGUMENT; }
  if ((reinterpret_cast<uintptr_t>(out0_bf16) & 7u) || (reinterpret_cast<uintptr_t>(out1_bf16) & 7u)) {
    set_error("instnorm: misaligned bf16 output"); return GC_BAD_ARGUMENT;
  }
  InstNormParams p{x, gz, u, out0, out1, (uint16_t*)out0_bf16, (uint16_t*)out1_bf16, N, HW, C, 0, eps, slope, act};
  static const char* names[3] = {"gc::instnorm_kernel<0>", "gc::instnorm_kernel<1>", "gc::instnorm_kernel<2>"};
  // up to 256 pixels: a wave per (sample, 64 or 16 channels), its pixels in registers (bf16 tensors; fp32 ones up to 16
  // pixels, or in the forward pass, where one tensor is held)
  static const bool wave_off = getenv("GC_NO_WAVE_NORM") != nullptr;
  const bool few = HW <= 16, mid = HW <= 4 * kCachePix, many = HW <= 16 * kCachePix;
  if (!wave_off && many && (dtype == GC_BF16 || mode == 0 || few)) {
    const int ql = mid ? 16 : 4;                  // channel quads per wave
    const long slabs = (long)N * ((C / 4 + ql - 1) / ql);
    const dim3 wgrid((unsigned)((slabs + kThreads / 64 - 1) / (kThreads / 64))), wblock(kThreads);
    ProfScope ps(names[mode], 0.0, stream);
#define GC_WAVE_LAUNCH(M_, T_, KC_, PL_) hipLaunchKernelGGL((instnorm_wave_kernel<M_, T_, KC_, PL_>), wgrid, wblock, 0, stream, p)
#define GC_WAVE_NORM(M_)                                                                                             \
    if (mode == M_) {                                                                                                  \
      if (dtype == GC_F32) {                                                                                           \
        if (few) GC_WAVE_LAUNCH(M_, float, 4, 4);                                                                      \
        else if (mid) GC_WAVE_LAUNCH(0, float, 16, 4);                                                                 \
        else GC_WAVE_LAUNCH(0, float, 16, 16);                                                                         \
      } else if (few) GC_WAVE_LAUNCH(M_, uint16_t, 4, 4);                                                              \
      else if (mid) GC_WAVE_LAUNCH(M_, uint16_t, 16, 4);                                                               \
      else GC_WAVE_LAUNCH(M_, uint16_t, 16, 16);                                                                       \
    }
    GC_WAVE_NORM(0) GC_WAVE_NORM(1) GC_WAVE_NORM(2)
#undef GC_WAVE_LAUNCH
#undef GC_WAVE_NORM
    return check_hip(hipGetLastError(), "instnorm launch");
  }
  while ((1 << p.qb_log2) < C / 4 && p.qb_log2 < 4) p.qb_log2++;
  // few samples with large images: narrower channel runs per workgroup (down to 16 channels) to fill the CUs
  while (p.qb_log2 > 2 && (long)N * ((C / 4 + (1 << p.qb_log2) - 1) >> p.qb_log2) < 512 && HW >= 1024) p.qb_log2--;
  const int QB = 1 << p.qb_log2;
  const int chunks = (C / 4 + QB - 1) / QB;
  const dim3 grid((unsigned)N * chunks), block(kThreads);
  ProfScope ps(names[mode], 0.0, stream);
  // register-resident slabs: at most kCachePix pixels per thread; fp32 tensors only in the forward pass (three fp32
  // tensors of 16 quads would need 192 registers)
  const bool cached = (long)HW * QB <= (long)kCachePix * kThreads && (dtype == GC_BF16 || mode == 0);
  if (dtype == GC_F32) {
    if (mode == 0 && cached) hipLaunchKernelGGL((instnorm_kernel<0, float, true>), grid, block, 0, stream, p);
    else if (mode == 0) hipLaunchKernelGGL((instnorm_kernel<0, float, false>), grid, block, 0, stream, p);
    else if (mode == 1) hipLaunchKernelGGL((instnorm_kernel<1, float, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((instnorm_kernel<2, float, false>), grid, block, 0, stream, p);
  } else if (cached) {
    if (mode == 0) hipLaunchKernelGGL((instnorm_kernel<0, uint16_t, true>), grid, block, 0, stream, p);
    else if (mode == 1) hipLaunchKernelGGL((instnorm_kernel<1, uint16_t, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((instnorm_kernel<2, uint16_t, true>), grid, block, 0, stream, p);
  } else {
    if (mode == 0) hipLaunchKernelGGL((instnorm_kernel<0, uint16_t, false>), grid, block, 0, stream, p);
    else if (mode == 1) hipLaunchKernelGGL((instnorm_kernel<1, uint16_t, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((instnorm_kernel<2, uint16_t, false>), grid, block, 0, stream, p);
  }
  return check_hip(hipGetLastError(), "instnorm launch");
}

}  // namespace gc

extern "C" {

int gc_instnorm_act_fwd(const void* x, void* z, void* z_bf16, int32_t N, int32_t HW, int32_t C, int32_t dtype, float eps, int32_t act,
                        float slope, void* stream) {
  return gc::launch_instnorm(0, dtype, x, nullptr, nullptr, z, nullptr, z_bf16, nullptr, N, HW, C, eps, slope, act, (hipStream_t)stream);
}

int gc_instnorm_act_bwd(const void* x, const void* gz, void* gx, void* gx_bf16, int32_t N, int32_t HW, int32_t C, int32_t dtype,
                        float eps, int32_t act, float slope, void* stream) {
  return gc::launch_instnorm(1, dtype, x, gz, nullptr, gx, nullptr, gx_bf16, nullptr, N, HW, C, eps, slope, act, (hipStream_t)stream);
}

int gc_instnorm_act_bwd_bwd(const void* x, const void* gz, const void* u, void* g_x, void* g_gz, void* g_x_bf16, void* g_gz_bf16,
                            int32_t N, int32_t HW, int32_t C, int32_t dtype, float eps, int32_t act, float slope, void* stream) {
  return gc::launch_instnorm(2, dtype, x, gz, u, g_x, g_gz, g_x_bf16, g_gz_bf16, N, HW, C, eps, slope, act, (hipStream_t)stream);
}

}  // extern "C"
