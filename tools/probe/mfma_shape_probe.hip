// Developer probe: v_mfma_f32_16x16x32_bf16 against v_mfma_f32_32x32x16_bf16 at the wave tile of the patch-resident
// convolution kernel (128 output channels x 64 pixels per wave, 128 accumulator registers), every operand fragment re-read
// from LDS with ds_read_b128 each K step of 32 as the kernel does (12 reads of 16 bytes per lane per step in both forms),
// random bf16 data, 4 waves per workgroup, 2 workgroups per CU.  Prints, per form: wall time (HIP events), TFLOP/s, the
// shader clock held inside the kernel (s_memtime ticks per s_memrealtime tick of 100 MHz) and cycles per MFMA.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_shape_probe.hip -o gpurun_out/mfma_shape_probe && gpurun_out/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int kSteps = 4096;        // K steps of 32 per wave
constexpr int kLdsBytes = 48 * 1024;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void probe_kernel(const uint32_t* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < kLdsBytes / 16; i += 256)
    reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(src)[(blockIdx.x % 8) * (kLdsBytes / 16) + i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // 12 fragments of 1 KiB per step (lane l reads bytes [16 l, 16 l + 16) of each: conflict-free), 3 steps' worth per wave
  const unsigned char* base = lds + wave * 12 * 1024 + lane * 16;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  float total = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < kSteps; s++) {
      bf16x8 a[8], b[4];
#pragma unroll
      for (int i = 0; i < 8; i++) a[i] = *reinterpret_cast<const bf16x8*>(base + i * 1024);
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = *reinterpret_cast<const bf16x8*>(base + (8 + j) * 1024);
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) total += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int s = 0; s < kSteps; s++) {
      bf16x8 a[4][2], b[2][2];      // [tile][K half of 16]
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int h = 0; h < 2; h++) a[i][h] = *reinterpret_cast<const bf16x8*>(base + (i * 2 + h) * 1024);
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int h = 0; h < 2; h++) b[j][h] = *reinterpret_cast<const bf16x8*>(base + (8 + j * 2 + h) * 1024);
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][h], b[j][h], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) total += acc[i][j][r];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  __builtin_amdgcn_s_waitcnt(0xC07F);
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = total;
  if (lane == 0) {
    stamps[((size_t)blockIdx.x * 4 + wave) * 2] = t1 - t0;
    stamps[((size_t)blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE>
void run(const char* name, const uint32_t* src, float* out, unsigned long long* stamps, int blocks) {
  auto kern = probe_kernel<SHAPE>;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; rep++) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), kLdsBytes, 0, src, out, stamps);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (int i = 0; i < blocks * 4; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
    const double flops = 2.0 * 128 * 64 * 32 * (double)kSteps * blocks * 4;
    const double ghz = cyc / real * 0.1;            // s_memrealtime ticks at 100 MHz
    const double mfmas = (SHAPE == 16 ? 32.0 : 16.0) * kSteps;
    printf("%-22s run %d: %8.3f ms  %7.1f TFLOP/s  shader clock %.3f GHz  %.2f cycles per MFMA (%.2f per 16x16x32-equivalent)\n", name, rep, ms,
           flops / ms / 1e9, ghz, cyc / (blocks * 4) / mfmas, cyc / (blocks * 4) / (32.0 * kSteps));
  }
}

int main() {
  int dev = 0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  const int blocks = prop.multiProcessorCount * 2;
  const size_t words = 8 * (size_t)kLdsBytes / 4;
  std::vector<uint32_t> h(words);
  srand(1);
  for (auto& w : h) {       // two random bf16 of magnitude about 1 with random sign and mantissa
    auto one = []() { return (uint32_t)(((rand() & 1) << 15) | ((126 + (rand() % 3)) << 7) | (rand() & 127)); };
    w = one() | (one() << 16);
  }
  uint32_t* src; float* out; unsigned long long* stamps;
  CHECK(hipMalloc(&src, words * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CHECK(hipMalloc(&stamps, (size_t)blocks * 8 * 8));
  CHECK(hipMemcpy(src, h.data(), words * 4, hipMemcpyHostToDevice));
  printf("%d CUs, %d workgroups of 4 waves, wave tile 128 x 64, %d K steps of 32, 12 ds_read_b128 per lane and step\n", prop.multiProcessorCount, blocks, kSteps);
  run<16>("mfma_f32_16x16x32_bf16", src, out, stamps, blocks);
  run<32>("mfma_f32_32x32x16_bf16", src, out, stamps, blocks);
  run<16>("mfma_f32_16x16x32_bf16", src, out, stamps, blocks);
  run<32>("mfma_f32_32x32x16_bf16", src, out, stamps, blocks);
  return 0;
}
