// The tail of BigGAN's self-attention block (BigGAN/model.py:97-105):  out = x + gamma * mixed.view(B, C, H, W), where
// `mixed` is the attention product [B, N, C] (N = H * W positions) that the reference REINTERPRETS, without a transpose, as
// a contiguous [B, C, H, W] tensor.  With channels-last activations that is a batched transpose fused with an axpy:
//
//   forward    out[b][p][c] = x[b][p][c] + gamma * m[b][c * N + p]            (m: the flat [N * C] buffer of a sample)
//   backward   dm[b][c * N + p] = gamma * g[b][p][c],   dgamma = sum over (b, p, c) of g[b][p][c] * m[b][c * N + p],   dx = g
//
// As tensor expressions these are a strided (non-vectorised) addcmul, a strided copy and two more sweeps per block and
// direction (0.9 TB/s on 400 MB).  Here a workgroup owns a tile of 64 positions x 64 channels: it reads the m (or g) tile
// with 16-byte loads along its contiguous axis, turns it through LDS (rows padded to 66 elements: the column reads of a
// wave fall on different banks) and writes along the other tensor's contiguous axis with 16-byte stores.  dgamma: one
// partial sum per workgroup, added in a fixed order by a second launch.
#include "common.h"

namespace gc {
namespace {

constexpr int kT = 64;            // tile side
constexpr int kPitch = kT + 2;    // LDS row pitch in elements

template <typename T> struct Vec;                      // 16 bytes of T
template <> struct Vec<uint16_t> { static constexpr int n = 8; };
template <> struct Vec<float> { static constexpr int n = 4; };

template <typename T>
__device__ __forceinline__ void load_vec(const T* p, float (&v)[Vec<T>::n]) {
  if constexpr (sizeof(T) == 4) {
    const f32x4 q = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = q[i];
  } else {
    const u32x4 q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[2 * i] = __builtin_bit_cast(float, q[i] << 16);
      v[2 * i + 1] = __builtin_bit_cast(float, q[i] & 0xffff0000u);
    }
  }
}
template <typename T>
__device__ __forceinline__ void store_vec(T* p, const float (&v)[Vec<T>::n]) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
  } else {
    const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
    *reinterpret_cast<u32x4*>(p) = u32x4{lo[0], lo[1], hi[0], hi[1]};
  }
}

// grid: (N / 64, C / 64, B).  FWD: src = m ([C][N] per sample), other = x, dst = out ([N][C]).
// BWD: src = g ([N][C]), other = m ([C][N]), dst = dm ([C][N]); part: one partial of dgamma per workgroup.
template <typename T, bool FWD>
__global__ __launch_bounds__(kThreads) void attn_residual_kernel(const T* __restrict__ src, const T* __restrict__ other, T* __restrict__ dst,
                                                                  const float* __restrict__ gamma, float* __restrict__ part, int N, int C) {
  constexpr int V = Vec<T>::n, VPR = kT / V;             // vectors per tile row
  __shared__ float tile[kT * kPitch];
  __shared__ float red[kThreads / 64];
  const int p0 = blockIdx.x * kT, c0 = blockIdx.y * kT;
  const size_t sample = (size_t)blockIdx.z * N * C;
  const float gm = gamma[0];
  // the source tile's rows run along its contiguous axis: FWD rows = channels (contiguous positions), BWD rows = positions
  const int srow0 = FWD ? c0 : p0, scol0 = FWD ? p0 : c0, spitch = FWD ? N : C;
  for (int i = threadIdx.x; i < kT * VPR; i += kThreads) {
    const int r = i / VPR, v = i % VPR;
    float a[V];
    load_vec<T>(src + sample + (size_t)(srow0 + r) * spitch + scol0 + v * V, a);
#pragma unroll
    for (int e = 0; e < V; e++) tile[r * kPitch + v * V + e] = a[e];
  }
  __syncthreads();
  // the destination tile's rows: FWD positions (contiguous channels), BWD channels (contiguous positions); element (r, col)
  // of it is element (col, r) of the source tile
  const int drow0 = FWD ? p0 : c0, dcol0 = FWD ? c0 : p0, dpitch = FWD ? C : N;
  float dot = 0.f;
  for (int i = threadIdx.x; i < kT * VPR; i += kThreads) {
    const int r = i / VPR, v = i % VPR;
    const size_t off = sample + (size_t)(drow0 + r) * dpitch + dcol0 + v * V;
    float o[V], t[V];
    load_vec<T>(other + off, o);
#pragma unroll
    for (int e = 0; e < V; e++) t[e] = tile[(v * V + e) * kPitch + r];
    if (FWD) {
#pragma unroll
      for (int e = 0; e < V; e++) o[e] = o[e] + gm * t[e];
      store_vec<T>(dst + off, o);
    } else {
#pragma unroll
      for (int e = 0; e < V; e++) { dot += t[e] * o[e]; t[e] *= gm; }
      store_vec<T>(dst + off, t);
    }
  }
  if (!FWD) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) dot += __shfl_xor(dot, s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < kThreads / 64; w++) s += red[w];
      part[((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
    }
  }
}

__global__ __launch_bounds__(kThreads) void attn_residual_dgamma_kernel(const float* part, float* dgamma, int n) {
  __shared__ float red[kThreads];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += kThreads) s += part[i];          // fixed order per thread
  red[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < kThreads; i++) t += red[i];
    dgamma[0] = t;
  }
}

int check(const char* who, int32_t B, int32_t N, int32_t C, int32_t dtype, const void* const* ptrs, int n) {
  if (B < 0 || N <= 0 || C <= 0 || N % kT != 0 || C % kT != 0 || B > 65535) {
    set_error("%s: N and C must be positive multiples of 64, B at most 65535 (got B=%d N=%d C=%d)", who, B, N, C);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  if (dtype != GC_F32 && dtype != GC_BF16) { set_error("%s: dtype must be GC_F32 or GC_BF16", who); return GC_BAD_ARGUMENT; }
  for (int i = 0; i < n; i++)
    if (ptrs[i] == nullptr || (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) != 0) {
      set_error("%s: every tensor must be non-null and 16-byte aligned", who);
      return GC_BAD_ARGUMENT;
    }
  return GC_OK;
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_attention_residual_fwd(const void* x, const void* mixed, const float* gamma, void* out, int32_t B, int32_t N, int32_t C,
                                         int32_t dtype, void* stream) {
  const void* ptrs[] = {x, mixed, gamma, out};
  if (int e = check("gc_attention_residual_fwd", B, N, C, dtype, ptrs, 4)) return e;
  if (B == 0) return GC_OK;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(N / kT, C / kT, B);
  ProfScope ps("gc::attn_residual_kernel<fwd>", 0.0, s);
  if (dtype == GC_F32)
    hipLaunchKernelGGL((attn_residual_kernel<float, true>), grid, dim3(kThreads), 0, s, (const float*)mixed, (const float*)x, (float*)out, gamma,
                       (float*)nullptr, N, C);
  else
    hipLaunchKernelGGL((attn_residual_kernel<uint16_t, true>), grid, dim3(kThreads), 0, s, (const uint16_t*)mixed, (const uint16_t*)x,
                       (uint16_t*)out, gamma, (float*)nullptr, N, C);
  return check_hip(hipGetLastError(), "attn_residual fwd launch");
}

extern "C" size_t gc_attention_residual_workspace_bytes(int32_t B, int32_t N, int32_t C) {
  if (B <= 0 || N <= 0 || C <= 0) return 0;
  return (size_t)B * (N / kT) * (C / kT) * sizeof(float);
}

extern "C" int gc_attention_residual_bwd(const void* g, const void* mixed, const float* gamma, void* dmixed, float* dgamma, void* workspace,
                                         size_t workspace_bytes, int32_t B, int32_t N, int32_t C, int32_t dtype, void* stream) {
  const void* ptrs[] = {g, mixed, gamma, dmixed, dgamma, workspace};
  if (int e = check("gc_attention_residual_bwd", B, N, C, dtype, ptrs, B == 0 ? 5 : 6)) return e;
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) return zero_fill(dgamma, sizeof(float), s);
  if (workspace_bytes < gc_attention_residual_workspace_bytes(B, N, C)) {
    set_error("gc_attention_residual_bwd: workspace of %zu bytes needed", gc_attention_residual_workspace_bytes(B, N, C));
    return GC_WORKSPACE_TOO_SMALL;
  }
  const dim3 grid(N / kT, C / kT, B);
  {
    ProfScope ps("gc::attn_residual_kernel<bwd>", 0.0, s);
    if (dtype == GC_F32)
      hipLaunchKernelGGL((attn_residual_kernel<float, false>), grid, dim3(kThreads), 0, s, (const float*)g, (const float*)mixed, (float*)dmixed,
                         gamma, (float*)workspace, N, C);
    else
      hipLaunchKernelGGL((attn_residual_kernel<uint16_t, false>), grid, dim3(kThreads), 0, s, (const uint16_t*)g, (const uint16_t*)mixed,
                         (uint16_t*)dmixed, gamma, (float*)workspace, N, C);
  }
  hipLaunchKernelGGL(attn_residual_dgamma_kernel, dim3(1), dim3(kThreads), 0, s, (const float*)workspace, dgamma, (int)(B * (N / kT) * (C / kT)));
  return check_hip(hipGetLastError(), "attn_residual bwd launch");
}
