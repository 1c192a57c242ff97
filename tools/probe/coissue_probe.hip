// Developer probe: do vector-ALU instructions run beside matrix instructions on one SIMD?  Each wave loops over 8
// independent v_mfma_f32_16x16x32_bf16 and NV independent vector instructions (v_fma_f32, v_pk_fma_f32 or v_exp_f32 chains
// on registers of their own), in one of two orders: interleaved in the instruction stream (one MFMA, then NV / 8 vector
// instructions) or grouped (8 MFMAs, then NV vector instructions: a second wave on the SIMD has to provide the overlap).
// WAVES waves per SIMD.  Prints cycles per loop iteration and SIMD; 8 MFMAs alone take 128.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/coissue_probe.hip -o tools/probe/_build/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

constexpr int kIters = 20000;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_exp_f32;  ORDER 0: interleaved, 1: grouped
template <int NV, int KIND, int ORDER>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[8];
  bf16x8 a, b;
  {
    const u32x4 q = {0x3f803f80u + (uint32_t)lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u};
    a = __builtin_bit_cast(bf16x8, q);
    b = __builtin_bit_cast(bf16x8, q);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x2 x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = f32x2{src[lane + i], src[lane + 8 + i]};
  const f32x2 m = {0.999f, 0.998f}, c = {1e-3f, 2e-3f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  auto valu = [&](int i) {
    if (KIND == 0) x[i & 7][0] = x[i & 7][0] * m[0] + c[0];
    else if (KIND == 1) x[i & 7] = x[i & 7] * m + c;
    else x[i & 7][0] = __builtin_amdgcn_exp2f(x[i & 7][0]);
  };
  for (int s = 0; s < kIters; s++) {
    if (ORDER == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < NV / 8; v++) valu(i * (NV / 8) + v);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; v++) valu(v);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float total = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) total += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + x[i][0] + x[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = total;
  if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = t1 - t0; stamps[1] = r1 - r0; }
}

template <int NV, int KIND, int ORDER>
void run(int waves_per_simd, const float* src, float* out, unsigned long long* stamps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int blocks = 256 * waves_per_simd;       // 256 CUs, one 4-wave workgroup per resident wave slot
  hipLaunchKernelGGL((probe<NV, KIND, ORDER>), dim3(blocks), dim3(256), 0, 0, src, out, stamps);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<NV, KIND, ORDER>), dim3(blocks), dim3(256), 0, 0, src, out, stamps);
  CHECK(hipEventRecord(e1));
  CHECK(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2];
  CHECK(hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / ((double)h[1] * 10.0);            // shader ticks per ns (the real-time counter runs at 100 MHz)
  const double cyc = ms * 1e6 * ghz / kIters / waves_per_simd;        // per iteration of ONE wave, as seen by the SIMD
  printf("kind %d order %d waves/SIMD %d  NV %3d: %8.3f ms  clock %.2f GHz  %7.1f cycles per wave-iteration on the SIMD (8 MFMAs = 128, NV vector ops = %d at 4 cycles)\n",
         KIND, ORDER, waves_per_simd, NV, ms, ghz, cyc, NV * 4);
}

int main() {
  float *src, *out;
  unsigned long long* stamps;
  CHECK(hipMalloc(&src, 4096));
  CHECK(hipMemset(src, 0, 4096));
  CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&stamps, 64));
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 0, 0>(w, src, out, stamps);
    run<16, 0, 0>(w, src, out, stamps);
    run<32, 0, 0>(w, src, out, stamps);
    run<64, 0, 0>(w, src, out, stamps);
    run<32, 0, 1>(w, src, out, stamps);
    run<64, 0, 1>(w, src, out, stamps);
    run<32, 1, 0>(w, src, out, stamps);
    run<32, 1, 1>(w, src, out, stamps);
    run<16, 2, 0>(w, src, out, stamps);
    run<32, 2, 0>(w, src, out, stamps);
    run<32, 2, 1>(w, src, out, stamps);
  }
  return 0;
}
