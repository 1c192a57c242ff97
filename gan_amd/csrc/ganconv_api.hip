// C ABI of libganconv.so (see include/ganconv.h): descriptor validation, tap tables, kernel selection, the
// weight repack and depthwise blur kernels, and the optional per-launch event timing.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "conv_gather.h"
#include "conv_wgrad.h"

namespace gc {

// ---- errors -------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return 0;
  set_error("%s: %s", what, hipGetErrorString(e));
  return GC_HIP_ERROR;
}

// ---- per-device set-up ----------------------------------------------------------------------------------
constexpr int kMaxDev = 64;
static std::mutex g_attr_mu;
static std::vector<const void*> g_attr_done[kMaxDev];

int ensure_dynamic_smem(const void* kernel, int bytes) {
  int dev = -1;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return GC_HIP_ERROR;
  if (dev < 0 || dev >= kMaxDev) { set_error("device index %d out of range", dev); return GC_BAD_ARGUMENT; }
  std::lock_guard<std::mutex> lk(g_attr_mu);
  for (const void* k : g_attr_done[dev])
    if (k == kernel) return GC_OK;
  if (check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), "hipFuncSetAttribute")) return GC_HIP_ERROR;
  g_attr_done[dev].push_back(kernel);
  return GC_OK;
}

int device_cu_count() {
  static std::atomic<int> cus[kMaxDev];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
  int n = cus[dev].load(std::memory_order_relaxed);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n, std::memory_order_relaxed);
  }
  return n;
}

// ---- profiling ----------------------------------------------------------------------------------------
struct ProfRec {
  std::string name;
  double flops;
  hipEvent_t e0, e1;
};
static std::atomic<bool> g_patch_kernel{true};
bool patch_kernel_allowed() { return g_patch_kernel.load(std::memory_order_relaxed); }
static std::atomic<bool> g_split_reduction{true};
bool split_reduction_allowed() { return g_split_reduction.load(std::memory_order_relaxed); }
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(const char* name, double flops, hipStream_t s) : slot(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r{name, flops, nullptr, nullptr};
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
  (void)hipEventRecord(r.e0, s);
  g_prof.push_back(r);
  slot = (int)g_prof.size() - 1;
}
ProfScope::~ProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_prof[slot].e1, stream);
}

// ---- weight repack -------------------------------------------------------------------------------------
// dst[plane][row][pitch] (bf16), a row being [tap][colp] followed by zeros up to `pitch`, from fp32 w[co][ci][r][s]
// with strides.  transpose == 0: row = co, col = ci (forward image); transpose == 1: row = ci, col = co
// (data-gradient image).
struct PackJob {
  const float* w;
  uint16_t* dst;
  int64_t s_co, s_ci, s_r, s_s;
  int32_t rows, cols, colp, pitch, kh, kw, transpose, planes;
  float alpha;
  const float* div;          // null, or one float in device memory: the image holds (alpha * w) / *div
  uint32_t block0;
  int32_t taps_contiguous;   // the kh * kw taps of a (co, ci) pair are consecutive floats: 1, or 4 when also 16-byte aligned in fours
};

// Workgroup `blk` of `nb` working on job q.  With contiguous taps a thread owns one (row, col) pair: it reads the
// pair's taps as one run of consecutive floats (every fetched byte is used; the element-per-thread form below reads
// 4 bytes out of every 64-byte run and fetches the tensor sixteen times) and stores one element per tap, neighbouring
// lanes to neighbouring addresses.
__device__ __forceinline__ void pack_job(const PackJob& q, uint32_t blk, uint32_t nb) {
  const size_t total = (size_t)q.rows * q.pitch;
  const int taps = q.kh * q.kw;
  // (a true division, as the tensor expression w / sigma rounds: the image equals the one packed from that tensor)
  const bool divided = q.div != nullptr;
  const float dv = divided ? *q.div : 1.f;
  auto scaled = [&](float v) { v *= q.alpha; return divided ? v / dv : v; };
  auto put = [&](size_t e, float v) {
    const uint16_t hi = float_to_bf16_bits(v);
    q.dst[e] = hi;
    if (q.planes == 2) q.dst[total + e] = float_to_bf16_bits(v - bf16_bits_to_float(hi));
  };
  if (q.taps_contiguous) {
    const size_t units = (size_t)q.rows * q.colp;
    for (size_t u = (size_t)blk * kThreads + threadIdx.x; u < units; u += (size_t)nb * kThreads) {
      const int row = (int)(u / q.colp), col = (int)(u - (size_t)row * q.colp);
      const size_t e0 = (size_t)row * q.pitch + col;
      if (col >= q.cols) {
        for (int tap = 0; tap < taps; tap++) put(e0 + (size_t)tap * q.colp, 0.f);
        continue;
      }
      const int co = q.transpose ? col : row, ci = q.transpose ? row : col;
      const float* src = q.w + co * q.s_co + ci * q.s_ci;
      if (q.taps_contiguous == 4) {
        for (int tap = 0; tap < taps; tap += 4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + tap);
#pragma unroll
          for (int i = 0; i < 4; i++) put(e0 + (size_t)(tap + i) * q.colp, scaled(v[i]));
        }
      } else {
        for (int tap = 0; tap < taps; tap++) put(e0 + (size_t)tap * q.colp, scaled(src[tap]));
      }
    }
    const int tail0 = taps * q.colp, tail = q.pitch - tail0;       // zeros behind the last tap of every row
    for (size_t u = (size_t)blk * kThreads + threadIdx.x; u < (size_t)q.rows * tail; u += (size_t)nb * kThreads) {
      const int row = (int)(u / tail);
      put((size_t)row * q.pitch + tail0 + (u - (size_t)row * tail), 0.f);
    }
    return;
  }
  for (size_t e = (size_t)blk * kThreads + threadIdx.x; e < total; e += (size_t)nb * kThreads) {
    const int k = (int)(e % q.pitch);
    const int row = (int)(e / q.pitch);
    const int col = k % q.colp, tap = k / q.colp;
    float v = 0.f;
    if (col < q.cols && tap < taps) {
      const int co = q.transpose ? col : row, ci = q.transpose ? row : col;
      v = scaled(q.w[co * q.s_co + ci * q.s_ci + (tap / q.kw) * q.s_r + (tap % q.kw) * q.s_s]);
    }
    put(e, v);
  }
}

__global__ __launch_bounds__(kThreads) void pack_weight_kernel(const PackJob q) { pack_job(q, blockIdx.x, gridDim.x); }

// Several weights in one launch (after an optimizer step): job j owns blocks [block0[j], block0[j + 1]).
constexpr int kMaxPackJobs = 24;
struct PackTable {
  int32_t n;
  uint32_t nblocks;
  PackJob job[kMaxPackJobs];
};
__global__ __launch_bounds__(kThreads) void pack_weights_kernel(const PackTable t) {
  int j = 0;
  for (int i = 1; i < t.n; i++)
    if (blockIdx.x >= t.job[i].block0) j = i;
  const PackJob& q = t.job[j];
  const uint32_t nb = (j + 1 < t.n ? t.job[j + 1].block0 : t.nblocks) - q.block0;
  pack_job(q, blockIdx.x - q.block0, nb);
}

// ---- depthwise 3x3 binomial blur, NHWC ----------------------------------------------------------------------
// [1 2 1] x [1 2 1] / 16 with zero padding, separable: a thread owns one column w of a strip of kBlurRows rows for a group
// of V channels (16-byte accesses: V = 4 fp32 or 8 bf16); per row it loads the pixel and its two horizontal neighbours
// (which its neighbour lanes load as well: L1 hits), forms the horizontal sum, and combines the last three horizontal
// sums from registers.  Reads of a strip: (kBlurRows + 2) / kBlurRows of the tensor; consecutive lanes run along
// (w, channel group), i.e. along memory.
constexpr int kBlurRows = 16;

template <typename T, int V>
__device__ __forceinline__ void blur_load(const T* x, size_t off, bool ok, float (&v)[V]) {
  if (!ok) {
#pragma unroll
    for (int i = 0; i < V; i++) v[i] = 0.f;
    return;
  }
  if constexpr (sizeof(T) == 4) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(x + off);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = q[i];
  } else if constexpr (V == 8) {
    const u32x4 q = *reinterpret_cast<const u32x4*>(x + off);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[2 * i] = __builtin_bit_cast(float, q[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, q[i] & 0xffff0000u);
    }
  } else {
    const u32x2 q = *reinterpret_cast<const u32x2*>(x + off);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      v[2 * i] = __builtin_bit_cast(float, q[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, q[i] & 0xffff0000u);
    }
  }
}

// MASK 1: the stored value is multiplied by 1 where mask > 0 and by `slope` elsewhere (mask: a tensor like y).
// MASK 2: every value READ is multiplied that way (mask: a tensor like x) -- the adjoint of MASK 1.
template <typename T, int V, int MASK>
__global__ __launch_bounds__(kThreads) void dwblur3x3_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ mask, float slope,
                                                             int N, int H, int W, int CV, int strips, uint32_t total) {
  const uint32_t e = blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int cv = (int)(e % (uint32_t)CV);
  uint32_t r = e / (uint32_t)CV;
  const int w = (int)(r % (uint32_t)W);
  r /= (uint32_t)W;
  const int strip = (int)(r % (uint32_t)strips);
  const int n = (int)(r / (uint32_t)strips);
  const int h0 = strip * kBlurRows, h1 = min(H, h0 + kBlurRows);
  const size_t rowp = (size_t)W * CV * V;                                   // elements per image row
  const size_t base = (size_t)n * H * rowp + (size_t)w * CV * V + (size_t)cv * V;
  const bool lok = w > 0, rok = w + 1 < W;
  float prev[V], cur[V], nxt[V];
  auto hsum = [&](int h, float (&o)[V]) {
    if ((unsigned)h >= (unsigned)H) {
#pragma unroll
      for (int i = 0; i < V; i++) o[i] = 0.f;
      return;
    }
    const size_t off = base + (size_t)h * rowp;
    float a[V], c[V], b[V];
    blur_load<T, V>(x, off - (size_t)CV * V, lok, a);
    blur_load<T, V>(x, off, true, c);
    blur_load<T, V>(x, off + (size_t)CV * V, rok, b);
    if constexpr (MASK == 2) {
      float ma[V], mc[V], mb[V];
      blur_load<T, V>(mask, off - (size_t)CV * V, lok, ma);
      blur_load<T, V>(mask, off, true, mc);
      blur_load<T, V>(mask, off + (size_t)CV * V, rok, mb);
#pragma unroll
      for (int i = 0; i < V; i++) {
        a[i] = ma[i] > 0.f ? a[i] : a[i] * slope;
        c[i] = mc[i] > 0.f ? c[i] : c[i] * slope;
        b[i] = mb[i] > 0.f ? b[i] : b[i] * slope;
      }
    }
#pragma unroll
    for (int i = 0; i < V; i++) o[i] = a[i] + 2.f * c[i] + b[i];
  };
  hsum(h0 - 1, prev);
  hsum(h0, cur);
  for (int h = h0; h < h1; h++) {
    hsum(h + 1, nxt);
    float o[V];
#pragma unroll
    for (int i = 0; i < V; i++) o[i] = (prev[i] + 2.f * cur[i] + nxt[i]) * (1.f / 16.f);
    const size_t off = base + (size_t)h * rowp;
    if constexpr (MASK == 1) {
      float m[V];
      blur_load<T, V>(mask, off, true, m);
#pragma unroll
      for (int i = 0; i < V; i++) o[i] = m[i] > 0.f ? o[i] : o[i] * slope;
    }
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<f32x4*>(y + off) = f32x4{o[0], o[1], o[2], o[3]};
    } else if constexpr (V == 8) {
      const u32x2 lo = cvt_bf16x4(f32x4{o[0], o[1], o[2], o[3]}), hi = cvt_bf16x4(f32x4{o[4], o[5], o[6], o[7]});
      *reinterpret_cast<u32x4*>(y + off) = u32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
      *reinterpret_cast<u32x2*>(y + off) = cvt_bf16x4(f32x4{o[0], o[1], o[2], o[3]});
    }
#pragma unroll
    for (int i = 0; i < V; i++) { prev[i] = cur[i]; cur[i] = nxt[i]; }
  }
}

// ---- descriptor handling ------------------------------------------------------------------------------
static int validate(const gc_conv_desc* d) {
  if (d == nullptr) { set_error("null descriptor"); return GC_BAD_DESC; }
  if (d->N < 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) {
    set_error("non-positive extent in descriptor (N=%d H=%d W=%d Cin=%d Ho=%d Wo=%d Cout=%d)", d->N, d->H, d->W, d->Cin, d->Ho, d->Wo, d->Cout);
    return GC_BAD_DESC;
  }
  if (d->kh <= 0 || d->kw <= 0 || d->stride_h <= 0 || d->stride_w <= 0 || d->pad_h < 0 || d->pad_w < 0) {
    set_error("bad filter geometry (k=%dx%d stride=%dx%d pad=%dx%d)", d->kh, d->kw, d->stride_h, d->stride_w, d->pad_h, d->pad_w);
    return GC_BAD_DESC;
  }
  for (int i = 0; i < 4; i++)
    if (d->reserved[i] != 0) { set_error("reserved descriptor fields must be zero"); return GC_BAD_DESC; }
  if ((d->x_dtype != GC_F32 && d->x_dtype != GC_BF16) || (d->y_dtype != GC_F32 && d->y_dtype != GC_BF16)) {
    set_error("unknown dtype"); return GC_BAD_DESC;
  }
  if (d->precision != GC_PREC_BF16 && d->precision != GC_PREC_F32X3) { set_error("unknown precision"); return GC_BAD_DESC; }
  // the conv's own output size may not exceed (Ho, Wo); H may exceed the minimal size (transposed conv output_padding)
  const int ho_full = (d->H + 2 * d->pad_h - d->kh) / d->stride_h + 1, wo_full = (d->W + 2 * d->pad_w - d->kw) / d->stride_w + 1;
  if (d->H + 2 * d->pad_h < d->kh || d->W + 2 * d->pad_w < d->kw || ho_full != d->Ho || wo_full != d->Wo) {
    set_error("Ho/Wo (%d,%d) do not match the convolution of H/W (%d,%d): expected (%d,%d)", d->Ho, d->Wo, d->H, d->W, ho_full, wo_full);
    return GC_BAD_DESC;
  }
  if (d->precision == GC_PREC_F32X3 && (d->x_dtype != GC_F32 || d->y_dtype != GC_F32)) { set_error("GC_PREC_F32X3 needs fp32 activations"); return GC_UNSUPPORTED_GEOMETRY; }
  if (d->kh * d->kw > kMaxTaps) { set_error("filters larger than %d taps are not supported", kMaxTaps); return GC_UNSUPPORTED_GEOMETRY; }
  if (d->stride_h * d->stride_w > 4) { set_error("stride %dx%d is not supported", d->stride_h, d->stride_w); return GC_UNSUPPORTED_GEOMETRY; }
  if (d->H >= 32768 || d->W >= 32768) { set_error("spatial extent too large"); return GC_UNSUPPORTED_GEOMETRY; }
  if (d->kh > 127 || d->kw > 127 || d->pad_h > 127 || d->pad_w > 127) { set_error("filter or padding too large"); return GC_UNSUPPORTED_GEOMETRY; }
  if ((long)d->N * d->H * d->W >= (1l << 31) || (long)d->N * d->Ho * d->Wo >= (1l << 31)) { set_error("more than 2^31 pixels"); return GC_UNSUPPORTED_GEOMETRY; }
  if (d->x_cstride != 0 && d->x_cstride < d->x_coff + d->Cin) { set_error("x channel slice exceeds x_cstride"); return GC_BAD_DESC; }
  if (d->y_cstride != 0 && d->y_cstride < d->y_coff + d->Cout) { set_error("y channel slice exceeds y_cstride"); return GC_BAD_DESC; }
  if (d->x_coff < 0 || d->y_coff < 0) { set_error("negative channel offset"); return GC_BAD_DESC; }
  return GC_OK;
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int x_cs(const gc_conv_desc* d) { return d->x_cstride ? d->x_cstride : d->Cin; }
static inline int y_cs(const gc_conv_desc* d) { return d->y_cstride ? d->y_cstride : d->Cout; }
static inline int elems_per_chunk(int dtype) { return dtype == GC_BF16 ? 8 : 4; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Input / output dtype of a pass and whether the combination is one the kernels implement.
static int pass_dtypes(const gc_conv_desc* d, int pass, int* in_dt, int* out_dt) {
  *in_dt = pass == GC_PASS_FPROP ? d->x_dtype : d->y_dtype;
  *out_dt = pass == GC_PASS_FPROP ? d->y_dtype : d->x_dtype;
  if (pass == GC_PASS_WGRAD) {
    if (d->x_dtype != d->y_dtype) { set_error("wgrad needs x and gy of one dtype"); return GC_UNSUPPORTED_GEOMETRY; }
    return GC_OK;
  }
  return GC_OK;
}

// out *= mask > 0 ? 1 : slope over the channel slice of every pixel: the grad_mask stage for the kernels that do not
// apply it in their own epilogue.  One thread per 4 channels.
template <typename T>
__global__ __launch_bounds__(kThreads) void mask_scale_kernel(void* out, const void* mask, size_t total4, uint32_t oc4, int cstride, int coff,
                                                               float slope) {
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total4; e += (size_t)gridDim.x * kThreads) {
    const size_t pix = e / oc4;
    const size_t idx = pix * cstride + coff + (e - pix * oc4) * 4;
    f32x4 v = load4<T>(out, idx);
    const f32x4 m = load4<T>(mask, idx);
#pragma unroll
    for (int r = 0; r < 4; r++) v[r] = m[r] > 0.f ? v[r] : v[r] * slope;
    store4<T>(out, idx, v);
  }
}

static int run_gather_unmasked(GatherParams& p, GatherLaunch& l, int out_dt, hipStream_t stream);

static int run_gather(GatherParams& p, GatherLaunch& l, int out_dt, hipStream_t stream) {
  if (p.mask == nullptr) return run_gather_unmasked(p, l, out_dt, stream);
  if (!p.out_vec || p.OC % 4 != 0) {
    set_error("grad_mask needs an output channel slice of a multiple of 4 channels at an aligned offset");
    return GC_UNSUPPORTED_GEOMETRY;
  }
  const void* mask = p.mask;
  const float slope = p.mask_slope;
  int rc = run_gather_unmasked(p, l, out_dt, stream);     // clears p.mask when its kernel applied the mask itself
  if (rc != GC_OK || p.mask == nullptr) return rc;
  const size_t total4 = (size_t)p.N * p.OH * p.OW * (p.OC / 4);
  size_t blocks = (total4 + kThreads - 1) / kThreads;
  if (blocks > 256 * 16) blocks = 256 * 16;
  ProfScope ps("gc::mask_scale_kernel", 0.0, stream);
  if (out_dt == GC_F32)
    hipLaunchKernelGGL(mask_scale_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, p.out, mask, total4, (uint32_t)(p.OC / 4),
                       p.out_cstride, p.out_coff, slope);
  else
    hipLaunchKernelGGL(mask_scale_kernel<uint16_t>, dim3((unsigned)blocks), dim3(kThreads), 0, stream, p.out, mask, total4,
                       (uint32_t)(p.OC / 4), p.out_cstride, p.out_coff, slope);
  return check_hip(hipGetLastError(), "mask_scale launch");
}

static int run_gather_unmasked(GatherParams& p, GatherLaunch& l, int out_dt, hipStream_t stream) {
  p.out_f32 = out_dt == GC_F32;
  if (p.out_bf16 != nullptr && (out_dt != GC_F32 || !p.out_vec || p.OC % 4 != 0)) {
    set_error("out_bf16 needs an fp32 output whose channel slice is a multiple of 4 channels at a multiple of 4");
    return GC_UNSUPPORTED_GEOMETRY;
  }
  // per-sample output scales (a modulated convolution): the tiled and patch-resident kernels apply them
  const bool scaled = p.pre_scale != nullptr || p.post_scale != nullptr;
  if (scaled && p.OC % 4 != 0) { set_error("sample_scale / sample_scale_out need a multiple of 4 output channels"); return GC_UNSUPPORTED_GEOMETRY; }
  if (!scaled) {
  if (plan_dot(p, l)) return launch_dot(p, l, out_dt == GC_F32, stream);
  if (plan_outer(p, l)) {
    const int rc = launch_outer(p, l, out_dt == GC_F32, stream);    // applies grad_mask itself
    p.mask = nullptr;
    return rc;
  }
  if (plan_spread(p, l)) return launch_spread(p, l, out_dt == GC_F32, stream);
  if (plan_image(p, l)) {
    const bool masked = image_applies_mask(p);
    const int rc = launch_image(p, l, out_dt == GC_F32, stream);
    if (masked) p.mask = nullptr;
    return rc;
  }
  if (plan_point(p, l)) return launch_point(p, l, out_dt == GC_F32, stream);
  if (p.out_bf16 == nullptr && plan_scatter(p, l)) return launch_scatter(p, l, out_dt == GC_F32, stream);
  }
  l.big_tile_ok = true;
  plan_gather(p, l);
  for (int c = 0; c < p.tt.n_classes; c++) {
    p.tt.cls_div_ow[c] = make_fastdiv((uint32_t)p.tt.cls_ow[c]);
    p.tt.cls_div_oh[c] = make_fastdiv((uint32_t)p.tt.cls_oh[c]);
  }
  p.out_vec8 = p.out_vec && (reinterpret_cast<uintptr_t>(p.out) % 16 == 0) && (p.out_cstride % 8 == 0) && (p.out_coff % 8 == 0) &&
               (reinterpret_cast<uintptr_t>(p.bias) % 16 == 0) && (reinterpret_cast<uintptr_t>(p.out_bf16) % 16 == 0);
  // LDS-DMA path: buffer addressing with 32-bit offsets, so both tensors have to stay below 2 GiB
  p.in_bytes = (int64_t)p.N * p.IH * p.IW * p.in_cstride * 2;
  p.w_bytes = (int64_t)p.OC * p.row_pitch * 2;
  long min_off = 0;
  for (int e = 0; e < p.tt.cls_tap_start[p.tt.n_classes]; e++) {
    const long off = ((long)p.tt.dh[e] * p.IW + p.tt.dw[e]) * p.in_cstride * 2;
    if (off < min_off) min_off = off;
  }
  p.a_bias = (int32_t)(-min_off);
  const int64_t limit = (1ll << 31) - (1 << 20);
  const bool dma = l.in_bf16 && l.mode == kVecLoads && p.ksplit == 1 && (l.to == 256 || l.to == 128 || l.to == 64) &&
                   p.in_bytes + p.a_bias < limit && p.w_bytes < limit;
  if (!dma && !scaled && p.ksplit > 1 && p.split_ws != nullptr && l.in_bf16 && l.mode == kVecLoads && (l.to == 128 || l.to == 64) &&
      p.in_bytes + p.a_bias < limit && p.w_bytes < limit &&
      p.split_ws_bytes >= (size_t)p.n_pix_tiles * p.n_oc_tiles * l.to * 128 * sizeof(float) * 8) {
    // a small output planned as a split reduction with atomic adds: with scratch memory on offer the wave-specialised
    // kernel splits it instead (partial tiles added in a fixed order: faster, and reproducible)
    GatherParams q = p;
    q.ksplit = 1;
    if (plan_ws(q, l)) {
      p.ksplit = 1;
      const int rc = launch_ws(p, l, out_dt == GC_F32, stream);
      p.mask = nullptr;
      return rc;
    }
  }
  if (dma && patch_kernel_allowed() && plan_thin(p, l)) return launch_thin(p, l, out_dt == GC_F32, stream);
  if (dma && patch_kernel_allowed() && plan_patch(p, l)) return launch_patch(p, l, out_dt == GC_F32, stream);
  if (dma && !scaled && plan_ws(p, l)) {
    const int rc = launch_ws(p, l, out_dt == GC_F32, stream);    // applies grad_mask in its epilogue
    p.mask = nullptr;
    return rc;
  }
  if (dma) return launch_gather_dma(p, l, out_dt == GC_F32, stream);
  return launch_gather(p, l, stream);
}

static double conv_flops(const gc_conv_desc* d) {
  return 2.0 * d->N * d->Ho * d->Wo * (double)d->Cout * d->Cin * d->kh * d->kw;
}

__global__ __launch_bounds__(kThreads) void zero_fill_kernel(uint32_t* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (size_t)gridDim.x * kThreads) p[i] = 0u;
}

int zero_fill(void* ptr, size_t bytes, hipStream_t stream) {
  const size_t n = bytes / 4;
  if (n == 0) return GC_OK;
  size_t blocks = (n + kThreads - 1) / kThreads;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, (uint32_t*)ptr, n);
  return check_hip(hipGetLastError(), "zero_fill launch");
}

static void fill_epilogue(GatherParams& p, const gc_epilogue* ep) {
  p.bias = nullptr; p.ch_sum = nullptr; p.ch_sumsq = nullptr; p.out_bf16 = nullptr; p.mask = nullptr; p.act = GC_ACT_NONE; p.slope = 0.f;
  p.stat_rows = nullptr; p.stat_cap = 0; p.stat_written = nullptr; p.split_ws = nullptr; p.split_ws_bytes = 0;
  p.pre_scale = nullptr; p.post_scale = nullptr;
  if (ep != nullptr) {
    p.pre_scale = ep->sample_scale; p.post_scale = ep->sample_scale_out;
    p.bias = ep->bias; p.ch_sum = ep->ch_sum; p.ch_sumsq = ep->ch_sumsq; p.act = ep->act; p.slope = ep->slope;
    p.out_bf16 = (uint16_t*)ep->out_bf16; p.mask = ep->grad_mask;
    if (p.mask != nullptr) { p.mask_slope = ep->slope; p.slope = 0.f; }
    p.split_ws = ep->workspace; p.split_ws_bytes = ep->workspace != nullptr ? ep->workspace_bytes : 0;
    if (ep->stat_rows != nullptr) {
      p.stat_rows = ep->stat_rows; p.stat_cap = ep->stat_rows_capacity; p.stat_written = ep->stat_rows_written;
      *p.stat_written = 0;
    }
  }
}

static int check_epilogue(const gc_epilogue* ep) {
  if (ep == nullptr) return GC_OK;
  if (ep->act < GC_ACT_NONE || ep->act > GC_ACT_SIGMOID) { set_error("unknown activation %d", ep->act); return GC_BAD_ARGUMENT; }
  if (ep->grad_mask != nullptr) {
    if (ep->act != GC_ACT_NONE || ep->ch_sum != nullptr || ep->out_bf16 != nullptr) {
      set_error("grad_mask needs act == GC_ACT_NONE, no statistics and no out_bf16");
      return GC_BAD_ARGUMENT;
    }
    if (reinterpret_cast<uintptr_t>(ep->grad_mask) & 15u) { set_error("grad_mask must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  }
  if (reinterpret_cast<uintptr_t>(ep->out_bf16) & 7u) { set_error("out_bf16 must be 8-byte aligned"); return GC_BAD_ARGUMENT; }
  if (ep->sample_scale != nullptr || ep->sample_scale_out != nullptr) {
    if (ep->ch_sum != nullptr || ep->stat_rows != nullptr || ep->grad_mask != nullptr || ep->out_bf16 != nullptr) {
      set_error("sample_scale / sample_scale_out exclude statistics, grad_mask and out_bf16"); return GC_BAD_ARGUMENT;
    }
    if ((reinterpret_cast<uintptr_t>(ep->sample_scale) & 15u) || (reinterpret_cast<uintptr_t>(ep->sample_scale_out) & 15u)) {
      set_error("sample_scale / sample_scale_out must be 16-byte aligned"); return GC_BAD_ARGUMENT;
    }
  }
  if (ep->ch_sumsq != nullptr && ep->ch_sum == nullptr) { set_error("ch_sumsq needs ch_sum"); return GC_BAD_ARGUMENT; }
  if (ep->stat_rows != nullptr) {
    if (ep->stat_rows_written == nullptr || ep->stat_rows_capacity < 0 || (reinterpret_cast<uintptr_t>(ep->stat_rows) & 7u)) {
      set_error("stat_rows needs stat_rows_written, a capacity >= 0 and 8-byte alignment"); return GC_BAD_ARGUMENT;
    }
    if (ep->ch_sum != nullptr || ep->grad_mask != nullptr) { set_error("stat_rows excludes ch_sum and grad_mask"); return GC_BAD_ARGUMENT; }
  }
  return GC_OK;
}

// The data gradient of a conv whose window covers its whole (unpadded) input -- one output pixel per sample, e.g.
// the 4x4 -> 1x1 layers at both ends of the DCGAN nets -- does not depend on the stride.  Treating the stride as the
// filter size gives every input pixel its own class with exactly one tap instead of kh*kw mostly-masked ones.
static int dgrad_class_stride(const gc_conv_desc* d) {
  if (d->Ho == 1 && d->Wo == 1 && d->pad_h == 0 && d->pad_w == 0 && d->H == d->kh && d->W == d->kw && d->kh == d->kw &&
      d->kh * d->kw <= kMaxClasses)
    return d->kh;
  return d->stride_h;
}

static PackGeom geom_for(const gc_conv_desc* d, int pass) {
  if (pass == GC_PASS_FPROP) return pack_geom(d->Cin, d->kh * d->kw, true);
  return pack_geom(d->Cout, d->kh * d->kw, dgrad_class_stride(d) == 1);
}

}  // namespace gc

using namespace gc;

extern "C" {

int gc_version(void) { return GANCONV_VERSION; }

int gc_set_patch_kernel(int on) { return g_patch_kernel.exchange(on != 0) ? 1 : 0; }

int gc_set_split_reduction(int on) { return g_split_reduction.exchange(on != 0) ? 1 : 0; }

const char* gc_last_error(void) { return g_err; }

size_t gc_packed_weight_bytes(const gc_conv_desc* d, int pass) {
  if (validate(d) != GC_OK) return 0;
  const int planes = d->precision == GC_PREC_F32X3 ? 2 : 1;
  if (pass != GC_PASS_FPROP && pass != GC_PASS_DGRAD) {
    set_error("gc_packed_weight_bytes: pass must be GC_PASS_FPROP or GC_PASS_DGRAD");
    return 0;
  }
  const PackGeom g = geom_for(d, pass);
  return (size_t)planes * (pass == GC_PASS_FPROP ? d->Cout : d->Cin) * g.row_pitch * 2;
}

// Fills the job of one weight tensor; returns the number of workgroups that cover it once.
static size_t fill_pack_job(gc::PackJob& q, const gc_conv_desc* d, int pass, const float* w, const int64_t* st, float alpha, void* packed) {
  const gc::PackGeom g = gc::geom_for(d, pass);
  q.w = w; q.dst = (uint16_t*)packed;
  q.s_co = st[0]; q.s_ci = st[1]; q.s_r = st[2]; q.s_s = st[3];
  q.transpose = pass == GC_PASS_DGRAD;
  q.rows = q.transpose ? d->Cin : d->Cout; q.cols = q.transpose ? d->Cout : d->Cin;
  q.colp = g.icp; q.pitch = g.row_pitch; q.kh = d->kh; q.kw = d->kw;
  q.planes = d->precision == GC_PREC_F32X3 ? 2 : 1;
  q.alpha = alpha;
  q.div = nullptr;
  q.block0 = 0;
  const int taps = d->kh * d->kw;
  q.taps_contiguous = 0;
  if (st[3] == 1 && (d->kh == 1 || st[2] == d->kw) && q.pitch >= taps * q.colp) {
    q.taps_contiguous = 1;
    if (taps % 4 == 0 && st[0] % 4 == 0 && st[1] % 4 == 0 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0) q.taps_contiguous = 4;
  }
  const size_t work = q.taps_contiguous ? (size_t)q.rows * q.colp : (size_t)q.rows * q.pitch;
  return (work + gc::kThreads - 1) / gc::kThreads;
}

int gc_pack_weight(const gc_conv_desc* d, int pass, const float* w, const int64_t w_strides[4], float alpha, void* packed,
                   void* stream) {
  int rc = validate(d);
  if (rc != GC_OK) return rc;
  if (w == nullptr || packed == nullptr || w_strides == nullptr) { set_error("gc_pack_weight: null pointer"); return GC_BAD_ARGUMENT; }
  if (pass != GC_PASS_FPROP && pass != GC_PASS_DGRAD) { set_error("gc_pack_weight: bad pass"); return GC_BAD_ARGUMENT; }
  if (!aligned16(packed)) { set_error("gc_pack_weight: packed buffer must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  PackJob q;
  size_t blocks = fill_pack_job(q, d, pass, w, w_strides, alpha, packed);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  ProfScope ps("gc::pack_weight_kernel", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream, q);
  return check_hip(hipGetLastError(), "pack_weight launch");
}

static int pack_weights_run(int32_t n, const gc_conv_desc* descs, const int32_t* passes, const float* const* ws, const int64_t* w_strides,
                            const float* alphas, const float* const* divisors, void* const* packed, void* stream) {
  if (n < 0 || (n > 0 && (descs == nullptr || passes == nullptr || ws == nullptr || w_strides == nullptr || packed == nullptr))) {
    set_error("gc_pack_weights: null pointer"); return GC_BAD_ARGUMENT;
  }
  for (int base = 0; base < n; base += kMaxPackJobs) {
    PackTable t;
    memset(&t, 0, sizeof(t));
    t.n = n - base < kMaxPackJobs ? n - base : kMaxPackJobs;
    uint32_t blocks = 0;
    for (int j = 0; j < t.n; j++) {
      const gc_conv_desc* d = descs + base + j;
      const int pass = passes[base + j];
      int rc = validate(d);
      if (rc != GC_OK) return rc;
      if (pass != GC_PASS_FPROP && pass != GC_PASS_DGRAD) { set_error("gc_pack_weights: bad pass"); return GC_BAD_ARGUMENT; }
      if (ws[base + j] == nullptr || packed[base + j] == nullptr || !aligned16(packed[base + j])) {
        set_error("gc_pack_weights: null weight or null / misaligned packed buffer"); return GC_BAD_ARGUMENT;
      }
      PackJob& q = t.job[j];
      size_t nb = fill_pack_job(q, d, pass, ws[base + j], w_strides + 4 * (size_t)(base + j), alphas != nullptr ? alphas[base + j] : 1.f,
                                packed[base + j]);
      q.div = divisors != nullptr ? divisors[base + j] : nullptr;
      q.block0 = blocks;
      if (nb > 512) nb = 512;
      if (nb < 1) nb = 1;
      blocks += (uint32_t)nb;
    }
    t.nblocks = blocks;
    ProfScope ps("gc::pack_weights_kernel", 0.0, (hipStream_t)stream);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(kThreads), 0, (hipStream_t)stream, t);
    if (check_hip(hipGetLastError(), "pack_weights launch")) return GC_HIP_ERROR;
  }
  return GC_OK;
}

int gc_pack_weights(int32_t n, const gc_conv_desc* descs, const int32_t* passes, const float* const* ws, const int64_t* w_strides,
                    const float* alphas, void* const* packed, void* stream) {
  return pack_weights_run(n, descs, passes, ws, w_strides, alphas, nullptr, packed, stream);
}

int gc_pack_weights_div(int32_t n, const gc_conv_desc* descs, const int32_t* passes, const float* const* ws, const int64_t* w_strides,
                        const float* alphas, const float* const* divisors, void* const* packed, void* stream) {
  return pack_weights_run(n, descs, passes, ws, w_strides, alphas, divisors, packed, stream);
}

int gc_conv2d_fprop(const gc_conv_desc* d, const void* x, const void* packed_w, void* y, const gc_epilogue* ep, void* stream) {
  int rc = validate(d);
  if (rc != GC_OK) return rc;
  if ((rc = check_epilogue(ep)) != GC_OK) return rc;
  if (d->N == 0) return GC_OK;
  if (x == nullptr || packed_w == nullptr || y == nullptr) { set_error("gc_conv2d_fprop: null pointer"); return GC_BAD_ARGUMENT; }
  if (!aligned16(packed_w)) { set_error("gc_conv2d_fprop: packed weights must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  GatherParams p;
  memset(&p, 0, sizeof(p));
  p.in = x; p.wp = (const uint16_t*)packed_w; p.out = y;
  fill_epilogue(p, ep);
  const PackGeom g = geom_for(d, GC_PASS_FPROP);
  p.N = d->N; p.IH = d->H; p.IW = d->W; p.IC = d->Cin; p.ICp = g.icp; p.row_pitch = g.row_pitch;
  p.OH = d->Ho; p.OW = d->Wo; p.OC = d->Cout; p.T = d->kh * d->kw;
  p.a = d->stride_h; p.b = 1;
  if (d->stride_h != d->stride_w) { set_error("anisotropic stride is not supported"); return GC_UNSUPPORTED_GEOMETRY; }
  p.in_cstride = x_cs(d); p.in_coff = d->x_coff; p.out_cstride = y_cs(d); p.out_coff = d->y_coff;
  p.plane_stride = (int64_t)d->Cout * g.row_pitch;
  p.tt.n_classes = 1;
  p.tt.cls_tap_start[0] = 0;
  for (int c = 1; c <= kMaxClasses; c++) p.tt.cls_tap_start[c] = p.T;
  p.tt.cls_ph[0] = p.tt.cls_pw[0] = 0; p.tt.cls_oh[0] = (int16_t)d->Ho; p.tt.cls_ow[0] = (int16_t)d->Wo;
  for (int r = 0; r < d->kh; r++)
    for (int s = 0; s < d->kw; s++) {
      const int i = r * d->kw + s;
      p.tt.dh[i] = (int8_t)(r - d->pad_h); p.tt.dw[i] = (int8_t)(s - d->pad_w); p.tt.wtap[i] = (uint8_t)i;
    }
  GatherLaunch l;
  l.in_bf16 = d->x_dtype == GC_BF16; l.x3 = d->precision == GC_PREC_F32X3;
  const int epc = elems_per_chunk(d->x_dtype);
  const bool vec = aligned16(x) && (p.in_cstride % epc == 0) && (p.in_coff % epc == 0) && (p.IC % epc == 0);
  l.mode = g.thin ? kThinInput : (vec ? kVecLoads : kElemLoads);
  for (p.icp_log2 = 0; (1 << p.icp_log2) < g.icp; p.icp_log2++) {}
  int in_dt, out_dt;
  if ((rc = pass_dtypes(d, GC_PASS_FPROP, &in_dt, &out_dt)) != GC_OK) return rc;
  p.out_vec = (reinterpret_cast<uintptr_t>(y) % (out_dt == GC_BF16 ? 8 : 16) == 0) && (p.out_cstride % 4 == 0) && (p.out_coff % 4 == 0);
  l.flops = conv_flops(d);
  return run_gather(p, l, out_dt, (hipStream_t)stream);
}

int gc_conv2d_dgrad(const gc_conv_desc* d, const void* gy, const void* packed_w, void* gx, const gc_epilogue* ep, void* stream) {
  int rc = validate(d);
  if (rc != GC_OK) return rc;
  if ((rc = check_epilogue(ep)) != GC_OK) return rc;
  if (d->N == 0) return GC_OK;
  if (gy == nullptr || packed_w == nullptr || gx == nullptr) { set_error("gc_conv2d_dgrad: null pointer"); return GC_BAD_ARGUMENT; }
  if (!aligned16(packed_w)) { set_error("gc_conv2d_dgrad: packed weights must be 16-byte aligned"); return GC_BAD_ARGUMENT; }
  if (d->stride_h != d->stride_w) { set_error("anisotropic stride is not supported"); return GC_UNSUPPORTED_GEOMETRY; }
  const int b = dgrad_class_stride(d);
  const PackGeom g = geom_for(d, GC_PASS_DGRAD);
  GatherParams p;
  memset(&p, 0, sizeof(p));
  p.in = gy; p.wp = (const uint16_t*)packed_w; p.out = gx;
  fill_epilogue(p, ep);
  p.N = d->N; p.IH = d->Ho; p.IW = d->Wo; p.IC = d->Cout; p.ICp = g.icp; p.row_pitch = g.row_pitch;
  p.OH = d->H; p.OW = d->W; p.OC = d->Cin; p.T = d->kh * d->kw;
  p.a = 1; p.b = b;
  p.in_cstride = y_cs(d); p.in_coff = d->y_coff; p.out_cstride = x_cs(d); p.out_coff = d->x_coff;
  p.plane_stride = (int64_t)d->Cin * g.row_pitch;
  // x pixel (h, w) receives gy[(h + pad - r)/b, (w + pad - s)/b] * w[r][s] for every tap with an exact quotient:
  // class (ph, pw) = (h mod b, w mod b) sees the taps r = (ph + pad) mod b, +b, ... at the fixed offset
  // dh = (ph + pad - r)/b from the class-local row h' = h / b.
  p.tt.n_classes = b * b;
  int e = 0;
  for (int ph = 0; ph < b; ph++)
    for (int pw = 0; pw < b; pw++) {
      const int c = ph * b + pw;
      p.tt.cls_tap_start[c] = e;
      p.tt.cls_ph[c] = (int8_t)ph; p.tt.cls_pw[c] = (int8_t)pw;
      p.tt.cls_oh[c] = (int16_t)((d->H - ph + b - 1) / b); p.tt.cls_ow[c] = (int16_t)((d->W - pw + b - 1) / b);
      for (int r = (ph + d->pad_h) % b; r < d->kh; r += b)
        for (int s = (pw + d->pad_w) % b; s < d->kw; s += b) {
          const int dh = (ph + d->pad_h - r) / b, dw = (pw + d->pad_w - s) / b;  // exact; may be negative
          p.tt.dh[e] = (int8_t)dh; p.tt.dw[e] = (int8_t)dw; p.tt.wtap[e] = (uint8_t)(r * d->kw + s);
          e++;
        }
    }
  for (int c = p.tt.n_classes; c <= kMaxClasses; c++) p.tt.cls_tap_start[c] = e;
  GatherLaunch l;
  l.in_bf16 = d->y_dtype == GC_BF16; l.x3 = d->precision == GC_PREC_F32X3;
  const int epc = elems_per_chunk(d->y_dtype);
  const bool vec = aligned16(gy) && (p.in_cstride % epc == 0) && (p.in_coff % epc == 0) && (p.IC % epc == 0);
  l.mode = g.thin ? kThinInput : (vec ? kVecLoads : kElemLoads);
  for (p.icp_log2 = 0; (1 << p.icp_log2) < g.icp; p.icp_log2++) {}
  int in_dt, out_dt;
  if ((rc = pass_dtypes(d, GC_PASS_DGRAD, &in_dt, &out_dt)) != GC_OK) return rc;
  p.out_vec = (reinterpret_cast<uintptr_t>(gx) % (out_dt == GC_BF16 ? 8 : 16) == 0) && (p.out_cstride % 4 == 0) && (p.out_coff % 4 == 0);
  l.flops = conv_flops(d);
  return run_gather(p, l, out_dt, (hipStream_t)stream);
}

static thread_local size_t g_rows_cap = 0;
}  // extern "C"
namespace gc {
size_t wgrad_rows_workspace_cap() { return g_rows_cap; }
}
extern "C" {
struct RowsCap {        // bounds the row-streaming plan's workspace for the calls made inside its scope (this thread)
  size_t prev;
  explicit RowsCap(size_t bytes) : prev(g_rows_cap) { g_rows_cap = bytes; }
  ~RowsCap() { g_rows_cap = prev; }
};

static void wgrad_setup(const gc_conv_desc* d, WgradParams& p, WgradLaunch& l, bool allow_dma) {
  memset(&p, 0, sizeof(p));
  p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout;
  p.kh = d->kh; p.kw = d->kw; p.sh = d->stride_h; p.sw = d->stride_w; p.ph = d->pad_h; p.pw = d->pad_w;
  p.x_cstride = x_cs(d); p.x_coff = d->x_coff; p.y_cstride = y_cs(d); p.y_coff = d->y_coff;
  p.K = d->N * d->Ho * d->Wo;
  p.div_wo = make_fastdiv((uint32_t)d->Wo); p.div_ho = make_fastdiv((uint32_t)d->Ho);
  // an fp32 x with a bf16 gy: planned as bf16 operands; only the image-layer kernel converts x itself (checked in wgrad_run)
  p.x_f32 = (d->x_dtype == GC_F32 && d->y_dtype == GC_BF16) ? 1 : 0;
  l.bf16 = d->y_dtype == GC_BF16 && (d->x_dtype == GC_BF16 || p.x_f32); l.x3 = d->precision == GC_PREC_F32X3;
  plan_wgrad(p, l, allow_dma);
}

// Weight partials [splits][T][Cout][Cin] (16-byte multiple), then the bias partials [splits][Cout] of the kernels that sum gy
// themselves.
static size_t wgrad_weight_part_bytes(const gc_conv_desc* d, const WgradParams& p) {
  const size_t b = (size_t)p.splits * d->kh * d->kw * d->Cout * (p.thin ? p.ICp : d->Cin) * sizeof(float);
  return (b + 15) / 16 * 16;
}
static size_t wgrad_need(const gc_conv_desc* d, const WgradParams& p) {
  return wgrad_weight_part_bytes(d, p) + (size_t)p.splits * d->Cout * sizeof(float);
}

// The LDS-DMA plan needs 16-byte aligned operands, which only the call knows: the advertised size covers both plans.
size_t gc_wgrad_workspace_bytes(const gc_conv_desc* d) {
  if (validate(d) != GC_OK) return 0;
  WgradParams p; WgradLaunch l;
  wgrad_setup(d, p, l, true);
  size_t need = wgrad_need(d, p);
  {   // the plans that replace the tiled one when their geometry (and the call's alignment) holds
    WgradParams q = p;
    if (plan_wgrad_rows(q, l, 1, &d->N) || plan_wgrad_score(q, l, true)) {
      const size_t other = wgrad_need(d, q);
      if (other > need) need = other;
    }
  }
  if (l.dma) {
    wgrad_setup(d, p, l, false);
    const size_t other = wgrad_need(d, p);
    if (other > need) need = other;
  }
  return need;
}

size_t gc_workspace_bytes(const gc_conv_desc* d, int pass) {
  if (validate(d) != GC_OK) return 0;
  if (pass == GC_PASS_WGRAD) return gc_wgrad_workspace_bytes(d);
  if (pass != GC_PASS_FPROP && pass != GC_PASS_DGRAD) { set_error("gc_workspace_bytes: unknown pass"); return 0; }
  const size_t out_elems = pass == GC_PASS_FPROP ? (size_t)d->N * d->Ho * d->Wo * d->Cout : (size_t)d->N * d->H * d->W * d->Cin;
  return out_elems * 4 * 32;        // partial fp32 tiles of up to 32 reduction splits
}

// Splits per segment for a plan made on the summed reduction length: every segment keeps the plan's k_per_split.
static void wgrad_split_segments(WgradParams& p, int32_t n_seg, const int32_t* Ns, int pix_per_sample) {
  p.n_seg = n_seg;
  int splits = 0;
  for (int g = 0; g < n_seg; g++) {
    p.seg_K[g] = Ns[g] * pix_per_sample;
    p.seg_split0[g] = splits;
    splits += (p.seg_K[g] + p.k_per_split - 1) / p.k_per_split;
  }
  p.splits = splits;
}

static int wgrad_run(const gc_conv_desc* d, int32_t n_seg, const void* const* xs, const void* const* gys, const int32_t* Ns, float* gw,
                     const int64_t gw_strides[4], float alpha, int accumulate, float* gbias, void* workspace, size_t workspace_bytes,
                     void* stream, const char* who) {
  int rc = validate(d);
  if (rc != GC_OK) return rc;
  bool al = true;
  for (int g = 0; g < n_seg; g++) {
    if (xs[g] == nullptr || gys[g] == nullptr) { set_error("%s: null pointer", who); return GC_BAD_ARGUMENT; }
    al = al && aligned16(xs[g]) && aligned16(gys[g]);
  }
  if (gw == nullptr || gw_strides == nullptr) { set_error("%s: null pointer", who); return GC_BAD_ARGUMENT; }
  WgradParams p; WgradLaunch l;
  wgrad_setup(d, p, l, al);
  const bool rows = plan_wgrad_rows(p, l, n_seg, Ns);
  if (!rows && n_seg > 1) wgrad_split_segments(p, n_seg, Ns, d->Ho * d->Wo);
  const bool score = !rows && n_seg == 1 && d->x_dtype == d->y_dtype && plan_wgrad_score(p, l, al);
  const size_t need = wgrad_need(d, p);
  if (workspace == nullptr || workspace_bytes < need) {
    set_error("%s: workspace of %zu bytes given, %zu needed", who, workspace_bytes, need);
    return GC_WORKSPACE_TOO_SMALL;
  }
  if (!aligned16(workspace)) { set_error("%s: workspace must be 16-byte aligned", who); return GC_BAD_ARGUMENT; }
  p.x = xs[0]; p.gy = gys[0]; p.part = (float*)workspace;
  p.bias_part = nullptr;
  for (int g = 0; g < n_seg && (n_seg > 1 || rows); g++) {
    p.seg_x[g] = xs[g]; p.seg_gy[g] = gys[g];
    p.seg_x_bytes[g] = (int64_t)Ns[g] * d->H * d->W * p.x_cstride * 2;
    p.seg_y_bytes[g] = (int64_t)Ns[g] * d->Ho * d->Wo * p.y_cstride * 2;
  }
  if (rows) p.n_seg = n_seg;
  if (d->x_dtype != d->y_dtype && !(p.x_f32 && p.image)) {
    set_error("%s: x and gy must have one dtype (an fp32 image with a bf16 gradient is taken in the image layer's geometry only)", who);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  const int epc = elems_per_chunk(d->y_dtype);
  const bool x_vec = al && (p.x_cstride % epc == 0) && (p.x_coff % epc == 0) && (p.Cin % epc == 0);
  const bool y_vec = al && (p.y_cstride % epc == 0) && (p.y_coff % epc == 0) && (p.Cout % epc == 0);
  // thin input: the template flag is about the gy side (the x side always takes element loads); otherwise it is about the
  // x side, and a gy side with an odd channel count (1- and 3-channel outputs) takes element loads inside that kernel
  l.aligned = p.thin ? y_vec : x_vec;
  p.y_elem = (!p.thin && x_vec && !y_vec) ? 1 : 0;
  l.flops = conv_flops(d);
  // an fp32 x beside a bf16 gy is converted by the image-layer kernel only: every other kernel would read it as bf16
  if (p.x_f32 && !wgrad_image_applies(p, l)) {
    set_error("%s: an fp32 x with a bf16 gy needs the image layer's geometry with 16-byte aligned, unsliced operands", who);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  // the LDS-DMA and row-streaming kernels sum gy next to the products (a fragment of ones): no pass of its own over gy
  const bool bias_fused = gbias != nullptr && (rows || l.dma) && d->N > 0;
  if (bias_fused) p.bias_part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + wgrad_weight_part_bytes(d, p));
  if (rows) rc = launch_wgrad_rows(p, l, (hipStream_t)stream);
  else if (score) rc = launch_wgrad_score(p, l, (hipStream_t)stream);
  else rc = l.dma ? launch_wgrad_dma(p, l, (hipStream_t)stream) : launch_wgrad(p, l, (hipStream_t)stream);
  if (rc != GC_OK) return rc;
  WgradReduceParams r;
  r.part = p.part; r.gw = gw;
  r.s_co = gw_strides[0]; r.s_ci = gw_strides[1]; r.s_r = gw_strides[2]; r.s_s = gw_strides[3];
  r.Cout = d->Cout; r.Cin = d->Cin; r.kh = d->kh; r.kw = d->kw; r.splits = d->N == 0 ? 0 : p.splits;
  r.alpha = alpha; r.accumulate = accumulate; r.thin = p.thin; r.ICp = p.ICp;
  r.bias_part = bias_fused ? p.bias_part : nullptr; r.gbias = gbias; r.main_blocks = 0;
  {
    ProfScope ps("gc::wgrad_reduce_kernel", 0.0, (hipStream_t)stream);
    if ((rc = launch_wgrad_reduce(r, (hipStream_t)stream)) != GC_OK) return rc;
  }
  if (gbias != nullptr && !bias_fused) {
    ProfScope ps("gc::colsum_kernel", 0.0, (hipStream_t)stream);
    if ((rc = launch_colsum(gys[0], gbias, p.K, d->Cout, p.y_cstride, p.y_coff, l.bf16, (hipStream_t)stream)) != GC_OK) return rc;
  }
  return GC_OK;
}

int gc_conv2d_wgrad(const gc_conv_desc* d, const void* x, const void* gy, float* gw, const int64_t gw_strides[4], float alpha,
                    int accumulate, float* gbias, void* workspace, size_t workspace_bytes, void* stream) {
  if (d == nullptr) { set_error("gc_conv2d_wgrad: null descriptor"); return GC_BAD_ARGUMENT; }
  const int32_t n = d->N;
  return wgrad_run(d, 1, &x, &gy, &n, gw, gw_strides, alpha, accumulate, gbias, workspace, workspace_bytes, stream, "gc_conv2d_wgrad");
}

int gc_wgrad_sums_bias(const gc_conv_desc* d, const void* x, const void* gy) {
  if (d == nullptr || validate(d) != GC_OK || d->N == 0) return 0;
  WgradParams p; WgradLaunch l;
  wgrad_setup(d, p, l, aligned16(x) && aligned16(gy));
  const bool rows = plan_wgrad_rows(p, l, 1, &d->N);
  return (rows || l.dma) ? 1 : 0;
}

static int multi_desc(const gc_conv_desc* d, int32_t n_seg, const int32_t* Ns, gc_conv_desc* total, const char* who) {
  if (d == nullptr || Ns == nullptr || n_seg < 1 || n_seg > kMaxWgradSegments) {
    set_error("%s: 1 to %d segments", who, kMaxWgradSegments);
    return GC_BAD_ARGUMENT;
  }
  *total = *d;
  long sum = 0;
  for (int g = 0; g < n_seg; g++) {
    if (Ns[g] <= 0) { set_error("%s: every segment needs at least one sample", who); return GC_BAD_ARGUMENT; }
    sum += Ns[g];
  }
  if (sum >= (1l << 31)) { set_error("%s: too many samples", who); return GC_UNSUPPORTED_GEOMETRY; }
  total->N = (int32_t)sum;
  return GC_OK;
}

size_t gc_wgrad_multi_workspace_bytes(const gc_conv_desc* d, int32_t n_seg, const int32_t* Ns) {
  gc_conv_desc t;
  if (multi_desc(d, n_seg, Ns, &t, "gc_wgrad_multi_workspace_bytes") != GC_OK || validate(&t) != GC_OK) return 0;
  size_t need = 0;
  for (int dma = 0; dma < 2; dma++) {      // the call knows the alignment, the query covers both plans
    WgradParams p; WgradLaunch l;
    wgrad_setup(&t, p, l, dma != 0);
    WgradParams q = p;
    if (!plan_wgrad_rows(p, l, n_seg, Ns) && n_seg > 1) wgrad_split_segments(p, n_seg, Ns, d->Ho * d->Wo);
    size_t b = wgrad_need(&t, p);
    if (b > need) need = b;
    if (n_seg == 1 && plan_wgrad_score(q, l, dma != 0) && (b = wgrad_need(&t, q)) > need) need = b;
  }
  return need;
}

int gc_conv2d_wgrad_multi(const gc_conv_desc* d, int32_t n_seg, const void* const* xs, const void* const* gys, const int32_t* Ns,
                          float* gw, const int64_t gw_strides[4], float alpha, int accumulate, void* workspace, size_t workspace_bytes,
                          void* stream) {
  gc_conv_desc t;
  int rc = multi_desc(d, n_seg, Ns, &t, "gc_conv2d_wgrad_multi");
  if (rc != GC_OK) return rc;
  if (xs == nullptr || gys == nullptr) { set_error("gc_conv2d_wgrad_multi: null pointer"); return GC_BAD_ARGUMENT; }
  return wgrad_run(&t, n_seg, xs, gys, Ns, gw, gw_strides, alpha, accumulate, nullptr, workspace, workspace_bytes, stream,
                   "gc_conv2d_wgrad_multi");
}

// ---- the entry point that allocates nothing -----------------------------------------------------------------------
// The scratch array lives in the code object: every GPU the library runs on holds its own copy at its own address.  In
// host code the array's NAME is only a handle for hipGetSymbolAddress; the address that kernels, memsets and copies may
// use is the one that call returns, looked up once per device with that device current.
static __device__ __align__(256) unsigned char scratch[GC_NOALLOC_SCRATCH_BYTES];
constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[kMaxDevices];

static int scratch_of_current_device(void** out) {
  int dev = -1;
  if (check_hip(hipGetDevice(&dev), "hipGetDevice")) return GC_HIP_ERROR;
  if (dev < 0 || dev >= kMaxDevices) { set_error("device index %d out of range", dev); return GC_BAD_ARGUMENT; }
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    void* ptr = nullptr;
    if (check_hip(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)")) return GC_HIP_ERROR;
    if (ptr == nullptr) { set_error("hipGetSymbolAddress returned a null scratch address"); return GC_HIP_ERROR; }
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return GC_OK;
}

int gc_conv2d_wgrad_noalloc(const gc_conv_desc* d, const void* x, const void* gy, float* gw, const int64_t gw_strides[4], float alpha,
                            int accumulate, float* gbias, void* stream) {
  int rc = validate(d);
  if (rc != GC_OK) return rc;
  RowsCap cap((size_t)GC_NOALLOC_SCRATCH_BYTES);
  const size_t need = gc_wgrad_workspace_bytes(d);
  if (need > (size_t)GC_NOALLOC_SCRATCH_BYTES) {
    set_error("gc_conv2d_wgrad_noalloc: the partial sums of this problem need %zu bytes, the built-in scratch holds %u; use "
              "gc_conv2d_wgrad with a workspace", need, (unsigned)GC_NOALLOC_SCRATCH_BYTES);
    return GC_WORKSPACE_TOO_SMALL;
  }
  void* ptr = nullptr;
  if ((rc = scratch_of_current_device(&ptr)) != GC_OK) return rc;
  // (nothing in the scratch has to start at zero: every partial slab that the reduction reads is written in full by
  // the weight-gradient kernel of the same call)
  return gc_conv2d_wgrad(d, x, gy, gw, gw_strides, alpha, accumulate, gbias, ptr, (size_t)GC_NOALLOC_SCRATCH_BYTES, stream);
}

int gc_noalloc_scratch(void** ptr, size_t* bytes) {
  if (ptr == nullptr || bytes == nullptr) { set_error("gc_noalloc_scratch: null pointer"); return GC_BAD_ARGUMENT; }
  *bytes = (size_t)GC_NOALLOC_SCRATCH_BYTES;
  return scratch_of_current_device(ptr);
}

}  // extern "C"

template <typename T, int V>
static void launch_blur(const void* x, void* y, const void* mask, float slope, int mode, int N, int H, int W, int C, int strips, size_t total,
                        hipStream_t stream) {
  const unsigned blocks = (unsigned)((total + kThreads - 1) / kThreads);
  if (mask != nullptr && mode == 2)
    hipLaunchKernelGGL((dwblur3x3_kernel<T, V, 2>), dim3(blocks), dim3(kThreads), 0, stream, (const T*)x, (T*)y, (const T*)mask, slope, N, H, W,
                       C / V, strips, (uint32_t)total);
  else if (mask != nullptr)
    hipLaunchKernelGGL((dwblur3x3_kernel<T, V, 1>), dim3(blocks), dim3(kThreads), 0, stream, (const T*)x, (T*)y, (const T*)mask, slope, N, H, W,
                       C / V, strips, (uint32_t)total);
  else
    hipLaunchKernelGGL((dwblur3x3_kernel<T, V, 0>), dim3(blocks), dim3(kThreads), 0, stream, (const T*)x, (T*)y, (const T*)nullptr, 0.f, N, H,
                       W, C / V, strips, (uint32_t)total);
}

static int blur_run(const void* x, void* y, const void* mask, float slope, int mode, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                    void* stream, const char* who) {
  if (N < 0 || H <= 0 || W <= 0 || C <= 0) { set_error("%s: bad shape", who); return GC_BAD_ARGUMENT; }
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("%s: unknown dtype", who); return GC_BAD_ARGUMENT; }
  if (N == 0) return GC_OK;
  if (x == nullptr || y == nullptr) { set_error("%s: null pointer", who); return GC_BAD_ARGUMENT; }
  if (C % 4 != 0 || !aligned16(x) || !aligned16(y) || !aligned16(mask)) {
    set_error("%s: channels must be a multiple of 4 and buffers 16-byte aligned (C=%d)", who, C);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  const int V = (dtype == GC_BF16 && C % 8 == 0) ? 8 : 4;
  const int strips = (H + kBlurRows - 1) / kBlurRows;
  const size_t total = (size_t)N * strips * W * (C / V);
  if (total >= (1ull << 32) - kThreads) { set_error("%s: tensor too large", who); return GC_UNSUPPORTED_GEOMETRY; }
  ProfScope ps(mask == nullptr ? "gc::dwblur3x3_kernel" : (mode == 2 ? "gc::dwblur3x3_premasked_kernel" : "gc::dwblur3x3_masked_kernel"), 0.0,
               (hipStream_t)stream);
  if (dtype == GC_F32) launch_blur<float, 4>(x, y, mask, slope, mode, N, H, W, C, strips, total, (hipStream_t)stream);
  else if (V == 4) launch_blur<uint16_t, 4>(x, y, mask, slope, mode, N, H, W, C, strips, total, (hipStream_t)stream);
  else launch_blur<uint16_t, 8>(x, y, mask, slope, mode, N, H, W, C, strips, total, (hipStream_t)stream);
  return check_hip(hipGetLastError(), "dwblur3x3 launch");
}

extern "C" {

int gc_dwblur3x3(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream) {
  return blur_run(x, y, nullptr, 0.f, 0, N, H, W, C, dtype, stream, "gc_dwblur3x3");
}

int gc_dwblur3x3_masked(const void* x, void* y, const void* mask, float slope, int mask_input, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t dtype, void* stream) {
  if (mask == nullptr) { set_error("gc_dwblur3x3_masked: null mask"); return GC_BAD_ARGUMENT; }
  return blur_run(x, y, mask, slope, mask_input ? 2 : 1, N, H, W, C, dtype, stream, "gc_dwblur3x3_masked");
}

int gc_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  return GC_OK;
}

int gc_profile_collect(int max_entries, char names[][96], int64_t* launches, double* total_ms, double* flops) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  int n = 0;
  for (auto& r : g_prof) {
    float ms = 0.f;
    if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      int i = 0;
      for (; i < n; i++)
        if (strncmp(names[i], r.name.c_str(), 95) == 0) break;
      if (i == n && n < max_entries) {
        strncpy(names[n], r.name.c_str(), 95);
        names[n][95] = 0;
        launches[n] = 0; total_ms[n] = 0.0; flops[n] = 0.0;
        n++;
      }
      if (i < n) { launches[i]++; total_ms[i] += ms; flops[i] += r.flops; }
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  return n;
}

}  // extern "C"
