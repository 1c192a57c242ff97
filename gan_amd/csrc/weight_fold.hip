// Weight-space prologues of the up-sampling convolutions: the 3 x 3 filter of "nearest x2, then 3 x 3 convolution" turned
// into the 4 x 4 filter of ONE stride-2 transposed convolution, and the adjoint map for the gradient.
//
//   w4[a][b] = mult * sum over (r, s) of M[a][r] * M[b][s] * w3[r][s]          (transpose_grad = 0)
//   g3[r][s] = mult * sum over (a, b) of M[a][r] * M[b][s] * g4[a][b]          (transpose_grad = 1)
//
// fold 0: M = [[1 0 0] [1 1 0] [0 1 1] [0 0 1]] / 2 -- StyleGAN's FusedUpsample, which pads the 3 x 3 weight to 5 x 5 and
//         averages its four one-step shifts (StyleGAN/model.py:396-402);
// fold 1: M = [[0 0 1] [0 1 1] [1 1 0] [1 0 0]]     -- the exact fold of nearest replication in front of a padding-1 3 x 3
//         convolution (BigGAN/model.py:116-120, StyleGAN/model.py:236-238): taps that land on the same input pixel add up.
// One thread per (i0, i1) filter: 9 floats in, 16 out (or 16 in, 9 out).  The 4 x 4 side is dense [n0][n1][4][4]; the 3 x 3
// side is addressed as i0 * stride0 + i1 * stride1 + 3 r + s, so it may be the transposed view of a [n1][n0][3][3] tensor.
#include "common.h"

namespace gc {
namespace {

__device__ __forceinline__ float fold_m(int fold, int a, int r) {
  if (fold == 0) return (r == a - 1 || r == a) ? 0.5f : 0.f;      // rows a - 1 and a of the unpadded filter, where they exist
  return (r == 2 - a || r == 3 - a) ? 1.f : 0.f;                  // fold 1: a = 0 -> r = 2; 1 -> 1, 2; 2 -> 0, 1; 3 -> 0
}

__global__ __launch_bounds__(kThreads) void upsample_weight_fold_kernel(const float* __restrict__ in, float* __restrict__ out, int n0, int n1,
                                                                        int64_t s0, int64_t s1, float mult, int transpose_grad, int fold) {
  const size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= (size_t)n0 * n1) return;
  const int i0 = (int)(i / n1), i1 = (int)(i - (size_t)i0 * n1);
  const size_t o3 = (size_t)i0 * s0 + (size_t)i1 * s1, o4 = i * 16;
  float m[4][3];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int r = 0; r < 3; r++) m[a][r] = fold_m(fold, a, r);
  if (!transpose_grad) {
    float w[3][3], t[4][3];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++) w[r][s] = in[o3 + r * 3 + s];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int s = 0; s < 3; s++) t[a][s] = m[a][0] * w[0][s] + m[a][1] * w[1][s] + m[a][2] * w[2][s];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) out[o4 + a * 4 + b] = mult * (t[a][0] * m[b][0] + t[a][1] * m[b][1] + t[a][2] * m[b][2]);
  } else {
    float g[4][4], t[3][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 4; b++) g[a][b] = in[o4 + a * 4 + b];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int b = 0; b < 4; b++) t[r][b] = m[0][r] * g[0][b] + m[1][r] * g[1][b] + m[2][r] * g[2][b] + m[3][r] * g[3][b];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int s = 0; s < 3; s++)
        out[o3 + r * 3 + s] = mult * (t[r][0] * m[0][s] + t[r][1] * m[1][s] + t[r][2] * m[2][s] + t[r][3] * m[3][s]);
  }
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_fused_upsample_weight(const float* w, float* w_out, int32_t n0, int32_t n1, int64_t stride0, int64_t stride1, float mult,
                                        int transpose_grad, int fold, void* stream) {
  if (n0 < 0 || n1 < 0 || (fold != 0 && fold != 1)) { set_error("gc_fused_upsample_weight: bad shape or fold"); return GC_BAD_ARGUMENT; }
  if ((size_t)n0 * n1 == 0) return GC_OK;
  if (w == nullptr || w_out == nullptr) { set_error("gc_fused_upsample_weight: null pointer"); return GC_BAD_ARGUMENT; }
  if (stride0 < 0 || stride1 < 0) { set_error("gc_fused_upsample_weight: negative stride"); return GC_BAD_ARGUMENT; }
  const size_t total = (size_t)n0 * n1;
  if (total >= (1ull << 31)) { set_error("gc_fused_upsample_weight: too many filters"); return GC_UNSUPPORTED_GEOMETRY; }
  ProfScope ps("gc::upsample_weight_fold_kernel", 0.0, (hipStream_t)stream);
  hipLaunchKernelGGL(upsample_weight_fold_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, (hipStream_t)stream, w,
                     w_out, n0, n1, stride0, stride1, mult, transpose_grad, fold);
  return check_hip(hipGetLastError(), "upsample_weight_fold launch");
}
