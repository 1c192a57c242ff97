// Fast path of the gather-GEMM convolution (forward and data gradient) for bf16 activations whose channel count
// is a multiple of 8: both operands go global -> LDS directly (LDS-DMA, `global_load_lds_dwordx4`), no register
// staging, no conversion, no LDS writes by the waves.  Same tile geometry, LDS image (128-byte rows, 16-byte
// chunks XOR-swizzled by (row >> 1) & 7) and MFMA loop as conv_gather_kernel; the swizzle is applied on the
// per-lane SOURCE address because an LDS-DMA wave-instruction writes 1 KiB linearly (8 rows x 128 bytes).
//
// A ring of STAGES LDS buffers keeps STAGES-1 reduction steps in flight: the loads of step k+STAGES-1 are issued
// before the MFMAs of step k, a counted `s_waitcnt vmcnt` retires only the oldest step, and one raw `s_barrier`
// per step both publishes the landed tile and frees the buffer that was just consumed.
// Pixels outside the image (zero padding) and channels past the end read from a small zero block instead.
#include <stdio.h>

#include "conv_gather.h"

namespace gc {

__device__ __attribute__((aligned(16))) const uint32_t g_zero_chunk[4] = {0u, 0u, 0u, 0u};

template <int TO, int TP, int NW>
struct DmaCfg {
  static constexpr int WO = (TO >= 64) ? 2 : 1;
  static constexpr int WP = NW / WO;
  static constexpr int FO = TO / WO / 16;
  static constexpr int FP = TP / WP / 16;
};

__device__ __forceinline__ int dswz(int row) { return (row >> 1) & 7; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <typename TOut, int TO, int TP, int STAGES, int NW>
__global__ __launch_bounds__(NW * 64) void conv_gather_dma_kernel(const GatherParams p) {
  using C = DmaCfg<TO, TP, NW>;
  constexpr int W_BYTES = TO * 128, A_BYTES = TP * 128;
  constexpr int STAGE = W_BYTES + A_BYTES;
  constexpr int AJ = TP / 8 / NW;  // activation wave-instructions per wave per step (8 rows each)
  constexpr int WJ = TO / 8 / NW;  // weight wave-instructions per wave per step
  constexpr int LOADS = AJ + WJ;  // LDS-DMA instructions per wave per step

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wo = wave / C::WP, wp_ = wave % C::WP;

  const uint32_t nblocks = (uint32_t)p.n_pix_tiles * (uint32_t)p.n_oc_tiles;
  const uint32_t lid = xcd_remap(blockIdx.x, nblocks);
  const int pix_tile = lid / p.n_oc_tiles;
  const int oc0 = (lid % p.n_oc_tiles) * TO;
  int cls = 0;
#pragma unroll
  for (int c = 1; c < kMaxClasses; c++)
    if (c < p.tt.n_classes && pix_tile >= p.tt.cls_tile_start[c]) cls = c;
  const int m0 = (pix_tile - p.tt.cls_tile_start[cls]) * TP;
  const int OHc = p.tt.cls_oh[cls], OWc = p.tt.cls_ow[cls];
  const int Mc = p.N * OHc * OWc;
  const int tap_begin = p.tt.cls_tap_start[cls], tap_end = p.tt.cls_tap_start[cls + 1];

  // ---- this lane's rows: wave-instruction j of this wave covers tile rows (wave*J + j)*8 .. +7, lane>>3 picks the row
  const int sub_row = lane >> 3, pos = lane & 7;
  int a_base[AJ], a_ih[AJ], a_iw[AJ], a_chunk[AJ];
  bool a_ok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; j++) {
    const int row = (wave * AJ + j) * 8 + sub_row;
    const int m = m0 + row;
    a_ok[j] = m < Mc;
    const int mm = a_ok[j] ? m : 0;
    const int ow = mm % OWc, tmp = mm / OWc;
    const int oh = tmp % OHc, n = tmp / OHc;
    a_base[j] = n * p.IH * p.IW;
    a_ih[j] = oh * p.a;
    a_iw[j] = ow * p.a;
    a_chunk[j] = (pos ^ dswz(row)) * 8;        // source chunk (elements) that lands at LDS position `pos` of this row
  }
  size_t w_off[WJ];
  int w_chunk[WJ];
  bool w_ok[WJ];
#pragma unroll
  for (int j = 0; j < WJ; j++) {
    const int row = (wave * WJ + j) * 8 + sub_row;
    const int oc = oc0 + row;
    w_ok[j] = oc < p.OC;
    w_off[j] = (size_t)(w_ok[j] ? oc : 0) * p.row_pitch;
    w_chunk[j] = (pos ^ dswz(row)) * 8;
  }
  const uint16_t* in = reinterpret_cast<const uint16_t*>(p.in);
  const void* zeros = g_zero_chunk;

  // The tap table is read through LDS inside the loop: an ordinary global load there would make the compiler
  // drain the in-flight LDS-DMA (vmcnt(0)) before using its result.
  int* tap_lds = reinterpret_cast<int*>(smem + STAGES * STAGE);
  if (t < kMaxTaps) tap_lds[t] = ((int)p.tt.dh[t] & 0xff) | (((int)p.tt.dw[t] & 0xff) << 8) | ((int)p.tt.wtap[t] << 16);
  __syncthreads();

  auto issue = [&](int stage, int tap, int c0) {
    const int e = __builtin_amdgcn_readfirstlane(tap_lds[tap]);
    const int dh = (int)(int8_t)(e & 0xff), dw = (int)(int8_t)((e >> 8) & 0xff), wt = e >> 16;
    char* wb = smem + stage * STAGE;
    char* ab = wb + W_BYTES;
#pragma unroll
    for (int j = 0; j < WJ; j++) {
      const void* src = w_ok[j] ? (const void*)(p.wp + w_off[j] + wt * p.ICp + c0 + w_chunk[j]) : zeros;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(wb + (wave * WJ + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < AJ; j++) {
      const int ih = a_ih[j] + dh, iw = a_iw[j] + dw;
      const bool ok = a_ok[j] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW && (c0 + a_chunk[j] < p.IC);
      const void* src = ok ? (const void*)(in + ((size_t)(a_base[j] + ih * p.IW + iw) * p.in_cstride + p.in_coff + c0 + a_chunk[j])) : zeros;
      __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(ab + (wave * AJ + j) * 1024), 16, 0, 0);
    }
  };

  f32x4 acc[C::FO][C::FP];
#pragma unroll
  for (int i = 0; i < C::FO; i++)
#pragma unroll
    for (int j = 0; j < C::FP; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int stage) {
    const char* wb = smem + stage * STAGE;
    const char* ab = wb + W_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int kc = ks * 4 + (lane >> 4);
      bf16x8 a[C::FO], b[C::FP];
#pragma unroll
      for (int f = 0; f < C::FO; f++) {
        const int row = wo * (TO / C::WO) + f * 16 + (lane & 15);
        a[f] = *reinterpret_cast<const bf16x8*>(wb + row * 128 + ((kc ^ dswz(row)) << 4));
      }
#pragma unroll
      for (int f = 0; f < C::FP; f++) {
        const int row = wp_ * (TP / C::WP) + f * 16 + (lane & 15);
        b[f] = *reinterpret_cast<const bf16x8*>(ab + row * 128 + ((kc ^ dswz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < C::FO; i++)
#pragma unroll
        for (int j = 0; j < C::FP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- main loop ---------------------------------------------------------------------------------------
  const int csteps = p.ICp / kBK;
  const int nsteps = (tap_end - tap_begin) * csteps;
  int tap = tap_begin, cs = 0;   // position of the NEXT step to issue
  int issued = 0;
  auto issue_next = [&]() {
    issue(issued % STAGES, tap, cs * kBK);
    issued++;
    if (++cs == csteps) { cs = 0; tap++; }
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; s++)
    if (issued < nsteps) issue_next();
  for (int it = 0; it < nsteps; it++) {
    // retire step `it`: at most the (issued - it - 1) younger steps may stay in flight
    const int younger = issued - it - 1;
    if (younger >= 2) wait_vmcnt<2 * LOADS>();
    else if (younger == 1) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of step `it` has landed; step it-1's buffer is free
    __builtin_amdgcn_sched_barrier(0);
    if (issued < nsteps) issue_next();            // goes into the buffer consumed at step it-1
    compute(it % STAGES);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  const bool vec_ok = p.out_vec != 0;
  float ssum[C::FO][4], ssq[C::FO][4];
#pragma unroll
  for (int i = 0; i < C::FO; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) ssum[i][r] = ssq[i][r] = 0.f;
  const bool stats = p.ch_sum != nullptr;
  TOut* out = reinterpret_cast<TOut*>(p.out);
#pragma unroll
  for (int j = 0; j < C::FP; j++) {
    const int m = m0 + wp_ * (TP / C::WP) + j * 16 + (lane & 15);
    const bool mok = m < Mc;
    const int mm = mok ? m : 0;
    const int ow = mm % OWc, tmp = mm / OWc;
    const int oh = tmp % OHc, n = tmp / OHc;
    const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
    TOut* dst = out + pix * p.out_cstride + p.out_coff;
#pragma unroll
    for (int i = 0; i < C::FO; i++) {
      const int oc = oc0 + wo * (TO / C::WO) + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float x = acc[i][j][r];
        if (p.bias != nullptr && oc + r < p.OC) x += p.bias[oc + r];
        x = apply_act(x, p.act, p.slope);
        if (sizeof(TOut) == 2) x = bf16_bits_to_float(float_to_bf16_bits(x));
        v[r] = x;
        if (stats && mok && oc + r < p.OC) { ssum[i][r] += x; ssq[i][r] += x * x; }
      }
      if (!mok) continue;
      if (vec_ok && oc + 3 < p.OC) {
        if (sizeof(TOut) == 4) *reinterpret_cast<f32x4*>(dst + oc) = f32x4{v[0], v[1], v[2], v[3]};
        else *reinterpret_cast<u32x2*>(dst + oc) = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]});
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oc + r < p.OC) {
            if (sizeof(TOut) == 4) reinterpret_cast<float*>(dst)[oc + r] = v[r];
            else reinterpret_cast<uint16_t*>(dst)[oc + r] = float_to_bf16_bits(v[r]);
          }
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int i = 0; i < C::FO; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float s = ssum[i][r], q = ssq[i][r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          s += __shfl_xor(s, d, 64);
          q += __shfl_xor(q, d, 64);
        }
        const int oc = oc0 + wo * (TO / C::WO) + i * 16 + (lane >> 4) * 4 + r;
        if ((lane & 15) == 0 && oc < p.OC) {
          atomicAdd(p.ch_sum + oc, s);
          if (p.ch_sumsq != nullptr) atomicAdd(p.ch_sumsq + oc, q);
        }
      }
  }
}

template <typename TOut, int TO, int TP, int STAGES, int NW>
static int launch_dma_one(const GatherParams& p, double flops, hipStream_t stream) {
  constexpr int smem = STAGES * (TO + TP) * 128 + kMaxTaps * 4;
  auto kern = conv_gather_dma_kernel<TOut, TO, TP, STAGES, NW>;
  static bool attr_set = false;
  if (smem > 64 * 1024 && !attr_set) {
    if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                  "hipFuncSetAttribute"))
      return GC_HIP_ERROR;
    attr_set = true;
  }
  const unsigned nblocks = (unsigned)p.n_pix_tiles * (unsigned)p.n_oc_tiles;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_gather_dma_kernel<%s, %d, %d, %d, %d>", sizeof(TOut) == 4 ? "float" : "unsigned short", TO, TP, STAGES, NW);
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NW * 64), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_gather_dma launch");
}

int launch_gather_dma(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t s) {
  if (p.n_pix_tiles == 0 || p.n_oc_tiles == 0) return GC_OK;
  if (l.to == 128 && l.tp == 256) return out_f32 ? launch_dma_one<float, 128, 256, 3, 8>(p, l.flops, s) : launch_dma_one<uint16_t, 128, 256, 3, 8>(p, l.flops, s);
  if (l.to == 128) return out_f32 ? launch_dma_one<float, 128, 128, 2, 4>(p, l.flops, s) : launch_dma_one<uint16_t, 128, 128, 2, 4>(p, l.flops, s);
  if (l.to == 64) return out_f32 ? launch_dma_one<float, 64, 128, 2, 4>(p, l.flops, s) : launch_dma_one<uint16_t, 64, 128, 2, 4>(p, l.flops, s);
  set_error("internal: no LDS-DMA variant for this tile");
  return GC_UNSUPPORTED_GEOMETRY;
}

}  // namespace gc
