// Per-sample channel scaling and its adjoint reduction, NHWC fp32 or bf16 storage, fp32 arithmetic:
//
//   scale    y[n, p, c] = x[n, p, c] * s[n, c]
//   rowdot   r[n, c]    = sum over the pixels p of a[n, p, c] * b[n, p, c]
//
// The two halves of a modulated convolution written without per-sample weights (StyleGAN2: w'[n, o, i, k] = s[n, i] *
// w[o, i, k], optionally normalised per (n, o)): scaling the INPUT channels by s[n, :] in front of an ordinary
// convolution and the OUTPUT channels by the normalisation factor behind it gives the same result as N grouped
// convolutions with N weight tensors.  An EXTENSION of this library: the reference repository's StyleGAN is version 1
// (AdaIN, StyleGAN/model.py:179-261) and has no modulated convolution.
//
// The pair is closed under differentiation: d scale / d x = scale(g, s), d scale / d s = rowdot(g, x);
// d rowdot / d a = scale(b, g), d rowdot / d b = scale(a, g).
//
// Both are memory-bound.  scale: 16-byte (fp32) / 8-byte (bf16) accesses of 4 channels, consecutive lanes along memory.
// rowdot: a workgroup owns 64 channels x one slab of a sample's pixels (16 channel quads x 16 pixel lanes), reduces its
// pixel lanes through LDS in a fixed order and writes one partial row; a second launch adds a sample's slabs in order
// (no float atomics: bitwise reproducible); with one slab per sample the first launch writes the result itself.
#include "common.h"

namespace gc {
namespace {

constexpr int kMQB = 16, kMPL = kThreads / kMQB;   // channel quads x pixel lanes of a workgroup

template <typename T>
__global__ __launch_bounds__(kThreads) void channel_scale_kernel(const void* x, const float* s, void* y, uint32_t total4, uint32_t c4,
                                                                 uint32_t hw_c4) {
  for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < total4; i += gridDim.x * kThreads) {
    const uint32_t n = i / hw_c4, q = i % c4;
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + ((size_t)n * c4 + q) * 4);
    store4<T>(y, (size_t)i * 4, load4<T>(x, (size_t)i * 4) * sv);
  }
}

// grid: samples x slabs x channel chunks
template <typename T>
__global__ __launch_bounds__(kThreads) void rowdot_partial_kernel(const void* a, const void* b, float* out, int HW, int C, int slabs,
                                                                  int slab_pix) {
  __shared__ f32x4 red[kThreads];
  const int t = threadIdx.x, tq = t & (kMQB - 1), tl = t >> 4;
  const int chunks = (C / 4 + kMQB - 1) / kMQB;
  int bid = blockIdx.x;
  const int chunk = bid % chunks; bid /= chunks;
  const int slab = bid % slabs, n = bid / slabs;
  const int c = (chunk * kMQB + tq) * 4;
  const int p0 = slab * slab_pix, p1 = min(HW, p0 + slab_pix);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C)
    for (int p = p0 + tl; p < p1; p += kMPL) {
      const size_t o = ((size_t)n * HW + p) * C + c;
      acc += load4<T>(a, o) * load4<T>(b, o);
    }
  red[tl * kMQB + tq] = acc;
  __syncthreads();
  if (tl == 0 && c < C) {
    f32x4 r = acc;
    for (int j = 1; j < kMPL; j++) r += red[j * kMQB + tq];
    *reinterpret_cast<f32x4*>(out + ((size_t)n * slabs + slab) * C + c) = r;
  }
}

// Backward sweep of a modulated convolution's fused output stage  out = post * act(pre * c + bias)  (act: none, ReLU or
// LeakyReLU, i.e. positively homogeneous, so that act'(u) * u = act(u)): from the incoming gradient g and the stored out
//   gc[n, p, c]  = pre[n, c] * m * post[n, c] * g[n, p, c],  m = 1 where post * out > 0 (the activation's argument was positive), else slope
//   dot[n, c]    = sum over p of g * out                      (= post * d L / d post, and the data term of d L / d pre)
//   sum[n, c]    = sum over p of m * post * g                 (the gradient of the pre-activation value: bias gradient per sample)
// Same workgroup shape and fixed-order reduction as rowdot_partial_kernel; the partial rows of dot and sum lie side by side.
template <typename T>
__global__ __launch_bounds__(kThreads) void mod_act_backward_kernel(const void* g, const void* out, const float* pre, const float* post, void* gc,
                                                                    float* part, int HW, int C, int slabs, int slab_pix, float slope) {
  __shared__ f32x4 red[2][kThreads];
  const int t = threadIdx.x, tq = t & (kMQB - 1), tl = t >> 4;
  const int chunks = (C / 4 + kMQB - 1) / kMQB;
  int bid = blockIdx.x;
  const int chunk = bid % chunks; bid /= chunks;
  const int slab = bid % slabs, n = bid / slabs;
  const int c = (chunk * kMQB + tq) * 4;
  const int p0 = slab * slab_pix, p1 = min(HW, p0 + slab_pix);
  f32x4 dot = {0.f, 0.f, 0.f, 0.f}, sum = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const f32x4 one = {1.f, 1.f, 1.f, 1.f};
    const f32x4 pr = pre != nullptr ? *reinterpret_cast<const f32x4*>(pre + (size_t)n * C + c) : one;
    const f32x4 po = post != nullptr ? *reinterpret_cast<const f32x4*>(post + (size_t)n * C + c) : one;
    for (int p = p0 + tl; p < p1; p += kMPL) {
      const size_t o = ((size_t)n * HW + p) * C + c;
      const f32x4 gv = load4<T>(g, o), ov = load4<T>(out, o);
      f32x4 gu;
#pragma unroll
      for (int r = 0; r < 4; r++) gu[r] = gv[r] * po[r] * (po[r] * ov[r] > 0.f ? 1.f : slope);
      dot += gv * ov;
      sum += gu;
      store4<T>(gc, o, gu * pr);
    }
  }
  red[0][tl * kMQB + tq] = dot;
  red[1][tl * kMQB + tq] = sum;
  __syncthreads();
  if (tl == 0 && c < C) {
    for (int j = 1; j < kMPL; j++) { dot += red[0][j * kMQB + tq]; sum += red[1][j * kMQB + tq]; }
    float* row = part + (((size_t)n * slabs + slab) * 2) * C;
    *reinterpret_cast<f32x4*>(row + c) = dot;
    *reinterpret_cast<f32x4*>(row + C + c) = sum;
  }
}

// part: [N][slabs][2][C] -> dot [N][C], sum [N][C]
__global__ __launch_bounds__(kThreads) void mod_act_backward_final_kernel(const float* part, float* dot, float* sum, int total, int C, int slabs) {
  const int i = blockIdx.x * kThreads + threadIdx.x;      // (n, c)
  if (i >= total) return;
  const int n = i / C, c = i - n * C;
  float d = 0.f, s = 0.f;
  for (int k = 0; k < slabs; k++) {
    const float* row = part + (((size_t)n * slabs + k) * 2) * C;
    d += row[c];
    s += row[C + c];
  }
  dot[i] = d;
  sum[i] = s;
}

__global__ __launch_bounds__(kThreads) void rowdot_final_kernel(const float* part, float* out, int total, int C, int slabs) {
  const int i = blockIdx.x * kThreads + threadIdx.x;      // (n, c)
  if (i >= total) return;
  const int n = i / C, c = i - n * C;
  float r = 0.f;
  for (int s = 0; s < slabs; s++) r += part[((size_t)n * slabs + s) * C + c];
  out[i] = r;
}

// slabs per sample: enough workgroups for about two per CU, a slab at least 64 pixels long
void rowdot_geometry(int32_t N, int32_t HW, int32_t C, int* slabs, int* slab_pix) {
  const int chunks = (C / 4 + kMQB - 1) / kMQB;
  long want = (512 + (long)N * chunks - 1) / ((long)N * chunks > 0 ? (long)N * chunks : 1);
  const long most = (HW + 63) / 64;
  if (want > most) want = most;
  if (want < 1) want = 1;
  *slab_pix = (int)((HW + want - 1) / want);
  *slabs = (HW + *slab_pix - 1) / *slab_pix;
}

bool bad_shape(int32_t N, int32_t HW, int32_t C, int32_t dtype, const char* who) {
  if (N < 0 || HW < 0 || C <= 0 || C % 4 != 0) { set_error("%s: N, HW >= 0 and C a positive multiple of 4 expected", who); return true; }
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("%s: dtype must be GC_F32 or GC_BF16", who); return true; }
  if ((long)N * HW * C >= (1l << 33)) { set_error("%s: tensor too large (N * HW * C must stay below 2^33)", who); return true; }
  return false;
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_channel_scale(const void* x, const float* s, void* y, int32_t N, int32_t HW, int32_t C, int32_t dtype, void* stream) {
  if (bad_shape(N, HW, C, dtype, "gc_channel_scale")) return GC_BAD_ARGUMENT;
  const size_t total = (size_t)N * HW * C;
  if (total == 0) return GC_OK;
  if (x == nullptr || s == nullptr || y == nullptr) { set_error("gc_channel_scale: null pointer"); return GC_BAD_ARGUMENT; }
  const size_t align = dtype == GC_F32 ? 15u : 7u;
  if (((uintptr_t)x & align) || ((uintptr_t)y & align) || ((uintptr_t)s & 15u)) {
    set_error("gc_channel_scale: x / y must be aligned to 4 elements, s to 16 bytes"); return GC_BAD_ARGUMENT;
  }
  const uint32_t total4 = (uint32_t)(total / 4), c4 = (uint32_t)(C / 4), hw_c4 = (uint32_t)HW * c4;
  unsigned blocks = (total4 + kThreads - 1) / kThreads;
  if (blocks > 8192) blocks = 8192;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps("gc::channel_scale_kernel", 0.0, st);
  if (dtype == GC_F32) hipLaunchKernelGGL(channel_scale_kernel<float>, dim3(blocks), dim3(kThreads), 0, st, x, s, y, total4, c4, hw_c4);
  else hipLaunchKernelGGL(channel_scale_kernel<uint16_t>, dim3(blocks), dim3(kThreads), 0, st, x, s, y, total4, c4, hw_c4);
  return check_hip(hipGetLastError(), "channel_scale launch");
}

extern "C" size_t gc_rowdot_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  int slabs, slab_pix;
  rowdot_geometry(N, HW, C, &slabs, &slab_pix);
  return slabs > 1 ? (size_t)N * slabs * C * sizeof(float) : 0;
}

extern "C" int gc_rowdot(const void* a, const void* b, float* out, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW,
                         int32_t C, int32_t dtype, void* stream) {
  if (bad_shape(N, HW, C, dtype, "gc_rowdot")) return GC_BAD_ARGUMENT;
  if ((size_t)N * C == 0) return GC_OK;
  if (out == nullptr || ((uintptr_t)out & 15u)) { set_error("gc_rowdot: out must be a 16-byte aligned pointer"); return GC_BAD_ARGUMENT; }
  hipStream_t st = (hipStream_t)stream;
  if (HW == 0) return zero_fill(out, (size_t)N * C * sizeof(float), st);
  if (a == nullptr || b == nullptr) { set_error("gc_rowdot: null pointer"); return GC_BAD_ARGUMENT; }
  const size_t align = dtype == GC_F32 ? 15u : 7u;
  if (((uintptr_t)a & align) || ((uintptr_t)b & align)) { set_error("gc_rowdot: a / b must be aligned to 4 elements"); return GC_BAD_ARGUMENT; }
  int slabs, slab_pix;
  rowdot_geometry(N, HW, C, &slabs, &slab_pix);
  const size_t need = gc_rowdot_workspace_bytes(N, HW, C);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 15u))) {
    set_error("gc_rowdot: workspace of %zu bytes (16-byte aligned) needed", need); return GC_BAD_ARGUMENT;
  }
  const int chunks = (C / 4 + kMQB - 1) / kMQB;
  const long blocks = (long)N * slabs * chunks;
  if (blocks >= (1l << 31)) { set_error("gc_rowdot: too many workgroups"); return GC_BAD_ARGUMENT; }
  float* first = slabs > 1 ? reinterpret_cast<float*>(workspace) : out;
  ProfScope ps("gc::rowdot_kernel", 0.0, st);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(rowdot_partial_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, st, a, b, first, HW, C, slabs, slab_pix);
  else
    hipLaunchKernelGGL(rowdot_partial_kernel<uint16_t>, dim3((unsigned)blocks), dim3(kThreads), 0, st, a, b, first, HW, C, slabs, slab_pix);
  if (slabs > 1) {
    const int total = N * C;
    hipLaunchKernelGGL(rowdot_final_kernel, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, first, out, total, C, slabs);
  }
  return check_hip(hipGetLastError(), "rowdot launch");
}

extern "C" size_t gc_modulated_act_backward_workspace_bytes(int32_t N, int32_t HW, int32_t C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  int slabs, slab_pix;
  rowdot_geometry(N, HW, C, &slabs, &slab_pix);
  return (size_t)N * slabs * 2 * C * sizeof(float);
}

extern "C" int gc_modulated_act_backward(const void* g, const void* out, const float* pre, const float* post, void* gc_out, float* dot,
                                         float* sum, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW, int32_t C,
                                         int32_t dtype, float slope, void* stream) {
  if (bad_shape(N, HW, C, dtype, "gc_modulated_act_backward")) return GC_BAD_ARGUMENT;
  if ((size_t)N * C == 0) return GC_OK;
  if (dot == nullptr || sum == nullptr) { set_error("gc_modulated_act_backward: null pointer"); return GC_BAD_ARGUMENT; }
  hipStream_t st = (hipStream_t)stream;
  if (HW == 0) {
    if (zero_fill(dot, (size_t)N * C * sizeof(float), st) != GC_OK) return GC_HIP_ERROR;
    return zero_fill(sum, (size_t)N * C * sizeof(float), st);
  }
  if (g == nullptr || out == nullptr || gc_out == nullptr) { set_error("gc_modulated_act_backward: null pointer"); return GC_BAD_ARGUMENT; }
  const size_t align = dtype == GC_F32 ? 15u : 7u;
  if (((uintptr_t)g & align) || ((uintptr_t)out & align) || ((uintptr_t)gc_out & align) || ((uintptr_t)pre & 15u) || ((uintptr_t)post & 15u)) {
    set_error("gc_modulated_act_backward: tensors must be aligned to 4 elements, pre / post to 16 bytes"); return GC_BAD_ARGUMENT;
  }
  int slabs, slab_pix;
  rowdot_geometry(N, HW, C, &slabs, &slab_pix);
  const size_t need = gc_modulated_act_backward_workspace_bytes(N, HW, C);
  if (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 15u)) {
    set_error("gc_modulated_act_backward: workspace of %zu bytes (16-byte aligned) needed", need); return GC_BAD_ARGUMENT;
  }
  const int chunks = (C / 4 + kMQB - 1) / kMQB;
  const long blocks = (long)N * slabs * chunks;
  if (blocks >= (1l << 31)) { set_error("gc_modulated_act_backward: too many workgroups"); return GC_BAD_ARGUMENT; }
  float* part = reinterpret_cast<float*>(workspace);
  ProfScope ps("gc::mod_act_backward_kernel", 0.0, st);
  if (dtype == GC_F32)
    hipLaunchKernelGGL(mod_act_backward_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), 0, st, g, out, pre, post, gc_out, part, HW, C,
                       slabs, slab_pix, slope);
  else
    hipLaunchKernelGGL(mod_act_backward_kernel<uint16_t>, dim3((unsigned)blocks), dim3(kThreads), 0, st, g, out, pre, post, gc_out, part, HW, C,
                       slabs, slab_pix, slope);
  const int total = N * C;
  hipLaunchKernelGGL(mod_act_backward_final_kernel, dim3((total + kThreads - 1) / kThreads), dim3(kThreads), 0, st, part, dot, sum, total, C, slabs);
  return check_hip(hipGetLastError(), "mod_act_backward launch");
}
