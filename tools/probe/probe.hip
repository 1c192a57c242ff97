// Runtime-sharing probe: a kernel built by the system hipcc, launched on a stream owned by PyTorch's
// bundled HIP runtime, on memory owned by PyTorch's caching allocator.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;

__global__ void scale_kernel(const float* x, float* y, float a, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}

// One wave: D[16x16] = A[16x32] * B[32x16], A given row-major [16][32], B given as Bt[16][32] (n-major), bf16 bits.
__global__ void mfma_probe(const short* A, const short* Bt, float* D) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; j++) {
    a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
    b[j] = Bt[(l & 15) * 32 + 8 * (l >> 4) + j];
  }
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

// Transposed LDS read probe: T[k=32][c=16] bf16 in LDS; produce A fragment A[c][k] via two tr reads per lane.
__global__ void tr_probe(const short* T, short* out /*[64][8]*/) {
  __shared__ __attribute__((aligned(16))) short lds[32 * 16];
  for (int i = threadIdx.x; i < 32 * 16; i += 64) lds[i] = T[i];
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  typedef __attribute__((address_space(3))) s4* lp;
  s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + (8 * g + q) * 16 + 4 * p));
  s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + (8 * g + 4 + q) * 16 + 4 * p));
  for (int j = 0; j < 4; j++) { out[l * 8 + j] = v0[j]; out[l * 8 + 4 + j] = v1[j]; }
}

extern "C" int probe_scale(const float* x, float* y, float a, int n, void* stream) {
  hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, a, n);
  return (int)hipGetLastError();
}
extern "C" int probe_mfma(const short* A, const short* Bt, float* D, void* stream) {
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bt, D);
  return (int)hipGetLastError();
}
extern "C" int probe_tr(const short* T, short* out, void* stream) {
  hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, T, out);
  return (int)hipGetLastError();
}
