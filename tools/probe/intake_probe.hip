// Probe: how many bytes per clock a CU takes in from L2 through (0) LDS-DMA, (1) vector loads + ds_write_b128 and
// (2) vector loads alone, with 4 loader waves per workgroup issuing 1 KiB wave-instructions of 128-byte rows (the
// access shape of the convolution kernels' tile loads).  Build: hipcc -O3 --offload-arch=gfx950 intake_probe.hip -o intake_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

constexpr int kLoads = 8;            // 1 KiB wave-instructions per wave per step (4 waves: 32 KiB per step)
constexpr int kStageBytes = 4 * kLoads * 1024;

template <int MODE, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void intake_kernel(const char* src, uint32_t region_bytes, int steps, uint32_t* sink,
                                                            unsigned long long* cycles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  // the workgroups that share an XCD (same blockIdx % 8) walk the same region, shifted: L2 hits after the first touch
  const char* base = src + (size_t)(blockIdx.x & 7) * region_bytes;
  const uint32_t start = ((blockIdx.x >> 3) * 4099u * 1024u) % region_bytes;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, (int)region_bytes, 0x00020000);
  // lane's offset inside a 1 KiB piece: 8 rows of 128 bytes, rows 256 bytes apart in memory (every other row of a tensor)
  const uint32_t lane_off = (uint32_t)(lane >> 3) * 256u + (uint32_t)(lane & 7) * 16u;
  uint32_t acc = 0;
  if (MODE == 0) {
    constexpr int STAGES = 4;
    int st = 0;
    for (int s = 0; s < steps; s++) {
      const uint32_t off = (start + (uint32_t)s * (kStageBytes * 2)) % (region_bytes - kStageBytes * 2);
#pragma unroll
      for (int j = 0; j < kLoads; j++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + st * kStageBytes + (wave * kLoads + j) * 1024), 16,
                                                 (int)(lane_off + (uint32_t)(wave * kLoads + j) * 2048u), (int)off, 0, 0);
      wait_vmcnt<(STAGES - 2) * kLoads>();
      st = st + 1 == STAGES ? 0 : st + 1;
    }
    wait_vmcnt<0>();
    acc = *reinterpret_cast<uint32_t*>(smem + t * 4);
  } else {
    u32x4 r[3][kLoads];
    auto load = [&](int slot, int s) {
      const uint32_t off = (start + (uint32_t)s * (kStageBytes * 2)) % (region_bytes - kStageBytes * 2);
#pragma unroll
      for (int j = 0; j < kLoads; j++)
        r[slot][j] = *reinterpret_cast<const u32x4*>(base + off + lane_off + (uint32_t)(wave * kLoads + j) * 2048u);
    };
    auto consume = [&](int slot, int s) {
#pragma unroll
      for (int j = 0; j < kLoads; j++) {
        if (MODE == 1) *reinterpret_cast<u32x4*>(smem + (s & 1) * kStageBytes + (wave * kLoads + j) * 1024 + lane * 16) = r[slot][j];
        else acc ^= r[slot][j][0] ^ r[slot][j][1] ^ r[slot][j][2] ^ r[slot][j][3];
      }
    };
    load(0, 0);
    load(1, 1);
    int s = 0;
    for (; s + 3 <= steps; s += 3) {
      load(2, s + 2); consume(0, s);
      load(0, s + 3); consume(1, s + 1);
      load(1, s + 4); consume(2, s + 2);
    }
    if (MODE == 1) { __syncthreads(); acc = *reinterpret_cast<uint32_t*>(smem + t * 4); }
  }
  if (acc == 0x9e3779b9u) sink[0] = acc;
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (t == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE, int NWAVES>
static void run(const char* name, const char* src, uint32_t region, int grid, int steps, uint32_t* sink, unsigned long long* dcyc) {
  const int smem = 4 * kStageBytes;
  hipFuncSetAttribute(reinterpret_cast<const void*>(intake_kernel<MODE, NWAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = MODE == 0 ? 4 * kStageBytes : (MODE == 1 ? 2 * kStageBytes : 1024);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((intake_kernel<MODE, NWAVES>), dim3(grid), dim3(NWAVES * 64), lds, 0, src, region, steps, sink, dcyc);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[4096];
    hipMemcpy(h, dcyc, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < grid; i++) cyc += (double)h[i];
    cyc /= grid;
    const double bytes = (double)grid * steps * (NWAVES * kLoads * 1024.0);
    if (rep == 2)
      printf("%-34s grid %4d waves %d: %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU  %5.1f B/clk per workgroup (in-kernel cycles)\n", name, grid,
             NWAVES, ms * 1e3, bytes / (ms * 1e-3) / 1e12, bytes / (ms * 1e-3) / 1e9 / 256.0, (double)steps * NWAVES * kLoads * 1024.0 / cyc);
  }
}

int main() {
  const uint32_t region = 4u << 20;     // 4 MiB per XCD: fits its L2
  char* src; uint32_t* sink; unsigned long long* dcyc;
  hipMalloc(&src, (size_t)region * 8);
  hipMemset(src, 1, (size_t)region * 8);
  hipMalloc(&sink, 64);
  hipMalloc(&dcyc, 4096 * 8);
  const int steps = 240;
  for (int grid : {256, 512}) {
    run<0, 4>("LDS-DMA (buffer_load ... lds)", src, region, grid, steps, sink, dcyc);
    run<1, 4>("global_load_dwordx4 + ds_write_b128", src, region, grid, steps, sink, dcyc);
    run<2, 4>("global_load_dwordx4 only", src, region, grid, steps, sink, dcyc);
  }
  // (8-wave variants were removed: the kernel's LDS allocation is sized for 4 waves, and the 8-wave register-staged
  //  variants wrote past it -- a GPU memory fault)
  return 0;
}
