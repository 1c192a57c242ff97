// Spectral norm of a weight matrix by power iteration (BigGAN/model.py:20-35, p = 2):
//     v = normalize(u W),   u' = normalize(v W^T),   sigma = u' W v^T
// for W [rows][cols] in row-major fp32 (a convolution weight flattened to [out, rest]) and the stored left vector u [rows],
// which is advanced in place.  With s = W v^T the last two are u' = s / max(|s|, eps) and sigma = |s|^2 / max(|s|, eps).
// HBM-bound: W is read twice per iteration, with 16-byte loads where the row length allows; every sum runs in a fixed
// order (partial sums per row range, added in order), so the result does not depend on timing.
//   sn_cols_kernel       part[split][col] = sum over the split's rows of u[row] * W[row][col]
//   sn_cols_finish       t = sum of the parts, scal[4 + b] = block b's share of |t|^2              (16 workgroups per matrix)
//   sn_rows_kernel       s[row] = <W[row], t> / max(|t|, eps), |t|^2 = the shares added in order   (one wave per row)
//   sn_rows_finish       u = s / max(|s|, eps), sigma = |s|^2 / max(|s|, eps)                    (one workgroup)
#include "common.h"

namespace gc {
namespace {

constexpr float kNormEps = 1e-12f;      // torch.nn.functional.normalize's default
constexpr int kMaxSplits = 64;
constexpr int kFinBlocks = 16;          // workgroups that add a matrix's column partials (one partial square each)
constexpr int kScalFloats = 4 + kFinBlocks;

__device__ __forceinline__ float block_sum(float v, float* red) {   // red: kThreads / 64 floats of LDS; all threads get the sum
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < kThreads / 64; w++) s += red[w];
  return s;
}

// (column block, split) of one matrix; a thread owns VEC consecutive columns
template <int VEC>
__device__ __forceinline__ void sn_cols(const float* w, const float* u, float* part, int rows, long cols, int rows_per_split, long col_block,
                                        int split) {
  const long c0 = (col_block * kThreads + threadIdx.x) * VEC;
  if (c0 >= cols) return;
  const int r0 = split * rows_per_split, r1 = min(rows, r0 + rows_per_split);
  float acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i++) acc[i] = 0.f;
  for (int r = r0; r < r1; r++) {
    const float ur = u[r];
    const float* src = w + (size_t)r * cols + c0;
    if (VEC == 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src);
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i] += ur * v[i];
    } else {
      acc[0] += ur * src[0];
    }
  }
  float* dst = part + (size_t)split * cols + c0;
#pragma unroll
  for (int i = 0; i < VEC; i++) dst[i] = acc[i];
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void sn_cols_kernel(const float* w, const float* u, float* part, int rows, long cols, int rows_per_split) {
  sn_cols<VEC>(w, u, part, rows, cols, rows_per_split, blockIdx.x, blockIdx.y);
}

// block `blk` of `nblk` takes the columns blk * kThreads + t, stride nblk * kThreads
__device__ __forceinline__ void sn_cols_fin(const float* part, float* t, float* scal, long cols, int splits, float* red, int blk, int nblk) {
  float sq = 0.f;
  for (long c = (long)blk * kThreads + threadIdx.x; c < cols; c += (long)nblk * kThreads) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four loads in flight; the order of the additions is fixed
    int k = 0;
    for (; k + 3 < splits; k += 4) {
      s0 += part[(size_t)k * cols + c];
      s1 += part[(size_t)(k + 1) * cols + c];
      s2 += part[(size_t)(k + 2) * cols + c];
      s3 += part[(size_t)(k + 3) * cols + c];
    }
    for (; k < splits; k++) s0 += part[(size_t)k * cols + c];
    const float s = (s0 + s1) + (s2 + s3);
    t[c] = s;
    sq += s * s;
  }
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) scal[4 + blk] = sq;
}

// 1 / max(|t|, eps) from the blocks' shares of |t|^2, added in block order
__device__ __forceinline__ float sn_inv_norm(const float* scal) {
  float sq = 0.f;
#pragma unroll
  for (int b = 0; b < kFinBlocks; b++) sq += scal[4 + b];
  return 1.f / fmaxf(sqrtf(sq), kNormEps);
}

__global__ __launch_bounds__(kThreads) void sn_cols_finish(const float* part, float* t, float* scal, long cols, int splits) {
  __shared__ float red[kThreads / 64];
  sn_cols_fin(part, t, scal, cols, splits, red, blockIdx.x, kFinBlocks);
}

// one wave per row
template <int VEC>
__device__ __forceinline__ void sn_rows(const float* w, const float* t, const float* scal, float* s, int rows, long cols, int row) {
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* src = w + (size_t)row * cols;
  float acc = 0.f;
  if (VEC == 4) {
    for (long c = lane * 4; c < cols; c += 64 * 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(src + c), b = *reinterpret_cast<const f32x4*>(t + c);
      acc += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    }
  } else {
    for (long c = lane; c < cols; c += 64) acc += src[c] * t[c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) s[row] = acc * sn_inv_norm(scal);
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void sn_rows_kernel(const float* w, const float* t, const float* scal, float* s, int rows, long cols) {
  sn_rows<VEC>(w, t, scal, s, rows, cols, blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6));
}

__device__ __forceinline__ void sn_rows_fin(const float* s, float* u, float* sigma, int rows, float* red) {
  float sq = 0.f;
  for (int r = threadIdx.x; r < rows; r += kThreads) sq += s[r] * s[r];
  sq = block_sum(sq, red);
  const float inv = 1.f / fmaxf(sqrtf(sq), kNormEps);
  for (int r = threadIdx.x; r < rows; r += kThreads) u[r] = s[r] * inv;
  if (threadIdx.x == 0) sigma[0] = sq * inv;
}

__global__ __launch_bounds__(kThreads) void sn_rows_finish(const float* s, float* u, float* sigma, int rows) {
  __shared__ float red[kThreads / 64];
  sn_rows_fin(s, u, sigma, rows, red);
}

// ---- every matrix of a network in one set of launches: a workgroup finds its matrix in the plan by its first block
__device__ __forceinline__ int plan_entry_of(const gc_spectral_plan* plan, int n, int block, bool rows_pass) {
  int e = 0;
  for (int i = 1; i < n; i++)
    if (block >= (rows_pass ? plan[i].row_block0 : plan[i].col_block0)) e = i;
  return e;
}

__global__ __launch_bounds__(kThreads) void sn_multi_cols_kernel(const gc_spectral_plan* plan, int n, float* ws) {
  const int e = plan_entry_of(plan, n, blockIdx.x, false);
  const gc_spectral_plan q = plan[e];
  const int local = blockIdx.x - q.col_block0;
  const int split = local / q.col_blocks, col_block = local - split * q.col_blocks;
  const float* w = static_cast<const float*>(q.w);
  const float* u = static_cast<const float*>(q.u);
  if (q.vec) sn_cols<4>(w, u, ws + q.ws_part, q.rows, q.cols, q.rows_per_split, col_block, split);
  else sn_cols<1>(w, u, ws + q.ws_part, q.rows, q.cols, q.rows_per_split, col_block, split);
}

__global__ __launch_bounds__(kThreads) void sn_multi_cols_finish(const gc_spectral_plan* plan, float* ws) {
  __shared__ float red[kThreads / 64];
  const gc_spectral_plan q = plan[blockIdx.x / kFinBlocks];
  sn_cols_fin(ws + q.ws_part, ws + q.ws_t, ws + q.ws_scal, q.cols, q.splits, red, blockIdx.x % kFinBlocks, kFinBlocks);
}

__global__ __launch_bounds__(kThreads) void sn_multi_rows_kernel(const gc_spectral_plan* plan, int n, float* ws) {
  const int e = plan_entry_of(plan, n, blockIdx.x, true);
  const gc_spectral_plan q = plan[e];
  const int row = (blockIdx.x - q.row_block0) * (kThreads / 64) + (threadIdx.x >> 6);
  const float* w = static_cast<const float*>(q.w);
  if (q.vec) sn_rows<4>(w, ws + q.ws_t, ws + q.ws_scal, ws + q.ws_s, q.rows, q.cols, row);
  else sn_rows<1>(w, ws + q.ws_t, ws + q.ws_scal, ws + q.ws_s, q.rows, q.cols, row);
}

__global__ __launch_bounds__(kThreads) void sn_multi_rows_finish(const gc_spectral_plan* plan, float* ws) {
  __shared__ float red[kThreads / 64];
  const gc_spectral_plan q = plan[blockIdx.x];
  sn_rows_fin(ws + q.ws_s, static_cast<float*>(q.u), static_cast<float*>(q.sigma), q.rows, red);
}

int splits_for(int rows, long cols, int vec) {
  const long col_blocks = (cols + (long)kThreads * vec - 1) / ((long)kThreads * vec);
  long want = (1024 + col_blocks - 1) / col_blocks;          // about four workgroups per CU
  if (want > rows / 8) want = rows / 8;                      // at least eight rows per split
  if (want > kMaxSplits) want = kMaxSplits;
  return want < 1 ? 1 : (int)want;
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" {

size_t gc_spectral_norm_workspace_bytes(int32_t rows, int64_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  // parts [splits][cols], t [cols], s [rows], the blocks' shares of |t|^2; each piece 16-byte aligned
  const size_t c4 = ((size_t)cols + 3) / 4 * 4, r4 = ((size_t)rows + 3) / 4 * 4;
  return ((size_t)kMaxSplits * c4 + c4 + r4 + kScalFloats) * sizeof(float);
}

int gc_spectral_norm_power_iteration(const float* w, int32_t rows, int64_t cols, float* u, float* sigma, int32_t n_iterations,
                                     void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (w == nullptr || u == nullptr || sigma == nullptr || rows <= 0 || cols <= 0 || n_iterations < 1) {
    set_error("gc_spectral_norm_power_iteration: null pointer, empty matrix or no iteration");
    return GC_BAD_ARGUMENT;
  }
  if (workspace == nullptr || workspace_bytes < gc_spectral_norm_workspace_bytes(rows, cols) ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) {
    set_error("gc_spectral_norm_power_iteration: workspace missing, too small (gc_spectral_norm_workspace_bytes) or not 16-byte aligned");
    return GC_BAD_ARGUMENT;
  }
  const bool vec = cols % 4 == 0 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0;
  const size_t c4 = ((size_t)cols + 3) / 4 * 4, r4 = ((size_t)rows + 3) / 4 * 4;
  float* part = static_cast<float*>(workspace);
  float* t = part + (size_t)kMaxSplits * c4;
  float* s = t + c4;
  float* scal = s + r4;
  const int v = vec ? 4 : 1;
  const int splits = splits_for(rows, cols, v);
  const int rows_per_split = (rows + splits - 1) / splits;
  const dim3 grid_c((unsigned)((cols + (long)kThreads * v - 1) / ((long)kThreads * v)), (unsigned)splits);
  const unsigned grid_r = (unsigned)((rows + kThreads / 64 - 1) / (kThreads / 64));
  for (int it = 0; it < n_iterations; it++) {
    {
      ProfScope ps("gc::sn_cols_kernel", 0.0, stream);
      if (vec) hipLaunchKernelGGL(sn_cols_kernel<4>, grid_c, dim3(kThreads), 0, stream, w, u, part, rows, (long)cols, rows_per_split);
      else hipLaunchKernelGGL(sn_cols_kernel<1>, grid_c, dim3(kThreads), 0, stream, w, u, part, rows, (long)cols, rows_per_split);
    }
    hipLaunchKernelGGL(sn_cols_finish, dim3(kFinBlocks), dim3(kThreads), 0, stream, part, t, scal, (long)cols, splits);
    {
      ProfScope ps("gc::sn_rows_kernel", 0.0, stream);
      if (vec) hipLaunchKernelGGL(sn_rows_kernel<4>, dim3(grid_r), dim3(kThreads), 0, stream, w, t, scal, s, rows, (long)cols);
      else hipLaunchKernelGGL(sn_rows_kernel<1>, dim3(grid_r), dim3(kThreads), 0, stream, w, t, scal, s, rows, (long)cols);
    }
    hipLaunchKernelGGL(sn_rows_finish, dim3(1), dim3(kThreads), 0, stream, s, u, sigma, rows);
  }
  return check_hip(hipGetLastError(), "spectral norm launch");
}


int gc_spectral_norm_multi_plan(const gc_spectral_job* jobs, int32_t n, gc_spectral_plan* plan, size_t* workspace_bytes, int32_t grids[2]) {
  if (jobs == nullptr || plan == nullptr || workspace_bytes == nullptr || grids == nullptr || n < 1) {
    set_error("gc_spectral_norm_multi_plan: null pointer or no matrix");
    return GC_BAD_ARGUMENT;
  }
  long col_blocks_total = 0, row_blocks_total = 0;
  size_t floats = 0;
  for (int i = 0; i < n; i++) {
    const gc_spectral_job& j = jobs[i];
    if (j.w == nullptr || j.u == nullptr || j.sigma == nullptr || j.rows <= 0 || j.cols <= 0) {
      set_error("gc_spectral_norm_multi_plan: matrix %d: null pointer or empty", i);
      return GC_BAD_ARGUMENT;
    }
    gc_spectral_plan& q = plan[i];
    q.w = j.w; q.u = j.u; q.sigma = j.sigma; q.rows = j.rows; q.cols = j.cols;
    q.vec = (j.cols % 4 == 0 && (reinterpret_cast<uintptr_t>(j.w) & 15u) == 0) ? 1 : 0;
    const int v = q.vec ? 4 : 1;
    q.splits = splits_for(j.rows, j.cols, v);
    q.rows_per_split = (j.rows + q.splits - 1) / q.splits;
    q.col_blocks = (int32_t)((j.cols + (long)kThreads * v - 1) / ((long)kThreads * v));
    q.col_block0 = (int32_t)col_blocks_total;
    q.row_block0 = (int32_t)row_blocks_total;
    col_blocks_total += (long)q.col_blocks * q.splits;
    row_blocks_total += (j.rows + kThreads / 64 - 1) / (kThreads / 64);
    const size_t c4 = ((size_t)j.cols + 3) / 4 * 4, r4 = ((size_t)j.rows + 3) / 4 * 4;
    q.ws_part = (int64_t)floats; floats += (size_t)q.splits * c4;
    q.ws_t = (int64_t)floats; floats += c4;
    q.ws_s = (int64_t)floats; floats += r4;
    q.ws_scal = (int64_t)floats; floats += kScalFloats;
  }
  if (col_blocks_total >= (1l << 31) || row_blocks_total >= (1l << 31)) {
    set_error("gc_spectral_norm_multi_plan: too many blocks");
    return GC_UNSUPPORTED_GEOMETRY;
  }
  *workspace_bytes = floats * sizeof(float);
  grids[0] = (int32_t)col_blocks_total;
  grids[1] = (int32_t)row_blocks_total;
  return GC_OK;
}

int gc_spectral_norm_multi_run(const gc_spectral_plan* plan_device, int32_t n, const int32_t grids[2], int32_t n_iterations, void* workspace,
                               void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (plan_device == nullptr || grids == nullptr || workspace == nullptr || n < 1 || n_iterations < 1 || grids[0] < 1 || grids[1] < 1 ||
      (reinterpret_cast<uintptr_t>(workspace) & 15u) != 0) {
    set_error("gc_spectral_norm_multi_run: null pointer, empty plan, no iteration or misaligned workspace");
    return GC_BAD_ARGUMENT;
  }
  float* ws = static_cast<float*>(workspace);
  for (int it = 0; it < n_iterations; it++) {
    {
      ProfScope ps("gc::sn_multi_cols_kernel", 0.0, stream);
      hipLaunchKernelGGL(sn_multi_cols_kernel, dim3((unsigned)grids[0]), dim3(kThreads), 0, stream, plan_device, n, ws);
    }
    hipLaunchKernelGGL(sn_multi_cols_finish, dim3((unsigned)n * kFinBlocks), dim3(kThreads), 0, stream, plan_device, ws);
    {
      ProfScope ps("gc::sn_multi_rows_kernel", 0.0, stream);
      hipLaunchKernelGGL(sn_multi_rows_kernel, dim3((unsigned)grids[1]), dim3(kThreads), 0, stream, plan_device, n, ws);
    }
    hipLaunchKernelGGL(sn_multi_rows_finish, dim3((unsigned)n), dim3(kThreads), 0, stream, plan_device, ws);
  }
  return check_hip(hipGetLastError(), "spectral norm (multi) launch");
}

}  // extern "C"
