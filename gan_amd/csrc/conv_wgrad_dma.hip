// Fast path of the weight gradient for bf16 operands whose channel counts are multiples of 8: both pixel-major
// tiles ([pixel][channel], the layout conv_wgrad_kernel keeps in LDS) are brought in by LDS-DMA
// (`buffer_load_dwordx4 ... lds`) through a ring of STAGES buffers, so several 32-pixel reduction steps are in flight
// per workgroup instead of one register-staged step.  The 32-byte-segment XOR swizzle of the LDS image is applied
// on each lane's SOURCE address (an LDS-DMA wave-instruction writes 1 KiB linearly).
//
// The 128 columns of a tile cover `tpt` consecutive filter taps of `cw` input channels each (cw * tpt = 128):
// one tap of 128 channels, or two taps of 64 for narrow layers, which then share the gy tile.
// The accumulators are kept transposed (rows = input channels) so that a lane owns 4 consecutive input channels
// of one output channel and the partial tile is written with 16-byte stores.
#include <stdio.h>

#include <mutex>

#include "conv_wgrad.h"

namespace gc {

namespace {

constexpr int kStepPix = 32;   // pixels per reduction step
constexpr int kTC = 128;       // tile columns

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int ROWB>
__device__ __forceinline__ int seg_swz(int k) {
  return ROWB >= 256 ? ((k & 3) | (((k >> 3) & 1) << 2)) : (((k >> 1) & 1) | (((k >> 3) & 1) << 1));
}

template <int N>
__device__ __forceinline__ void wait_vm() {   // s_waitcnt vmcnt(N) only (expcnt / lgkmcnt left at "no wait")
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

}  // namespace

// BIAS: the variant that also sums gy (a build of its own, so that the plain one keeps its registers and schedule)
template <int TO, int STAGES, bool BIAS>
__global__ __launch_bounds__(kThreads) void conv_wgrad_dma_kernel(const WgradParams p) {
  constexpr int ROWBY = TO * 2, ROWBX = kTC * 2;
  constexpr int Y_BYTES = kStepPix * ROWBY, X_BYTES = kStepPix * ROWBX;
  constexpr int STAGE = Y_BYTES + X_BYTES;
  constexpr int CPRY = ROWBY / 16, CPRX = ROWBX / 16;      // 16-byte chunks per row
  constexpr int RPIY = 64 / CPRY, RPIX = 64 / CPRX;        // rows per wave-instruction
  constexpr int YJ = kStepPix / RPIY / 4, XJ = kStepPix / RPIX / 4;   // wave-instructions per wave per step
  constexpr int LOADS = YJ + XJ;
  constexpr int FO = TO / 32, FC = kTC / 32;
  static_assert(FC == 4 && (FO == 8 || FO == 4 || FO == 2), "the fragment wait lists its registers explicitly");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wo = wave >> 1, wc = wave & 1;

  // (tile, tap group) fastest, split slowest, contiguous runs per XCD: one split's workgroups share their pixels
  const uint32_t ntiles = (uint32_t)p.n_co_tiles * (uint32_t)p.n_ci_tiles, ngroups = (uint32_t)p.n_tap_groups;
  uint32_t lid = xcd_remap(blockIdx.x, ntiles * ngroups * (uint32_t)p.splits);
  const int tile = (int)(lid % ntiles);
  lid /= ntiles;
  const int tap0 = (int)(lid % ngroups) * p.tpt;
  const int split = (int)(lid / ngroups);
  const int co0 = (tile / p.n_ci_tiles) * TO, ci0 = (tile % p.n_ci_tiles) * kTC;
  const int T = p.kh * p.kw;
  const WgradRange rg = wgrad_range(p, split);
  const int kb0 = rg.kb0, kend = rg.kend;

  // Buffer addressing: a lane whose source is padding, past the channel count or past the last pixel uses an
  // offset beyond the buffer's range and receives zeros.  Both tensors are below 2 GiB (checked by the planner).
  constexpr uint32_t kOutOfRange = 0x80000000u;
  const __amdgpu_buffer_rsrc_t y_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rg.gy), 0, (int)rg.y_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(rg.x), 0, (int)rg.x_bytes, 0x00020000);

  // ---- per-lane source geometry (fixed over the steps except for the pixel index)
  // gy: wave-instruction j of this wave covers tile rows (wave*YJ + j)*RPIY .. ; lane / CPRY picks the row
  const int y_rsub = lane / CPRY, y_pos = lane % CPRY;
  const int x_rsub = lane / CPRX, x_pos = lane % CPRX;
  uint32_t y_voff[YJ];   // byte offset of (tile row, source chunk) relative to the step's first pixel
#pragma unroll
  for (int j = 0; j < YJ; j++) {
    const int row = (wave * YJ + j) * RPIY + y_rsub;
    const int ch = (((y_pos >> 1) ^ seg_swz<ROWBY>(row)) << 1) | (y_pos & 1);
    const int c = co0 + ch * 8;
    y_voff[j] = c < p.Cout ? (uint32_t)((row * p.y_cstride + p.y_coff + c) * 2) : kOutOfRange;
  }
  uint32_t x_c2[XJ];
  int x_r[XJ], x_s[XJ], x_row[XJ];
#pragma unroll
  for (int j = 0; j < XJ; j++) {
    const int row = (wave * XJ + j) * RPIX + x_rsub;
    const int ch = (((x_pos >> 1) ^ seg_swz<ROWBX>(row)) << 1) | (x_pos & 1);
    const int col = ch * 8;
    const int tap = tap0 + (col >> p.cw_log2);
    const int ci = ci0 + (col & ((1 << p.cw_log2) - 1));
    const bool ok = tap < T && (col >> p.cw_log2) < p.tpt && ci < p.Cin;
    const int tp = ok ? tap : 0;
    x_r[j] = ok ? tp / p.kw - p.ph : -(1 << 20);   // a masked column fails the row check of every pixel
    x_s[j] = tp % p.kw - p.pw;
    x_c2[j] = (uint32_t)((p.x_coff + ci) * 2);
    x_row[j] = row;
  }
  const uint32_t x_cs2 = (uint32_t)p.x_cstride * 2u;

  auto issue = [&](int stage, int kb) {
    char* yb = smem + stage * STAGE;
    char* xb = yb + Y_BYTES;
    const uint32_t s_y = (uint32_t)kb * (uint32_t)p.y_cstride * 2u;   // pixels past the last one fall out of range
#pragma unroll
    for (int j = 0; j < YJ; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rs, (lds_ptr_t)(yb + (wave * YJ + j) * 1024), 16, (int)(y_voff[j] + s_y), 0, 0, 0);
#pragma unroll
    for (int j = 0; j < XJ; j++) {
      const uint32_t k = (uint32_t)(kb + x_row[j]);
      const uint32_t q1 = fd_div(k, p.div_wo);
      const int wo_ = (int)(k - q1 * p.div_wo.d);
      const uint32_t n = fd_div(q1, p.div_ho);
      const int ho_ = (int)(q1 - n * p.div_ho.d);
      const int ih = ho_ * p.sh + x_r[j], iw = wo_ * p.sw + x_s[j];
      const bool ok = ((unsigned)ih < (unsigned)p.H) & ((unsigned)iw < (unsigned)p.W);
      const uint32_t pix = (n * (uint32_t)p.H + (uint32_t)ih) * (uint32_t)p.W + (uint32_t)iw;   // n >= N lands out of range
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xb + (wave * XJ + j) * 1024), 16,
                                               (int)(ok ? pix * x_cs2 + x_c2[j] : kOutOfRange), 0, 0, 0);
    }
  };

  f32x4 acc[FO][FC];
#pragma unroll
  for (int i = 0; i < FO; i++)
#pragma unroll
    for (int j = 0; j < FC; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing read: lane (g = lane>>4, q = (lane>>2)&3, pp = lane&3) addresses row 8g+q (+4), columns 4pp..4pp+3
  // of a 16-channel segment and receives, for channel lane&15, the four rows 8g..8g+3 (+4): exactly k = 8g+j.
  // The reads are written as inline assembly: the compiler orders its own LDS reads behind every outstanding
  // LDS-DMA (vmcnt(0)), which would serialise the ring; the counted wait + barrier above already ordered them.
  const int k0 = 8 * (lane >> 4) + ((lane >> 2) & 3);
  const int sub = (lane & 3) * 8;
  const int fy = seg_swz<ROWBY>(k0), fx = seg_swz<ROWBX>(k0);
  // LDS byte address (within a stage) of this lane's first read of each fragment
  uint32_t ya[FO], xa[FC];
#pragma unroll
  for (int i = 0; i < FO; i++) ya[i] = (uint32_t)(k0 * ROWBY + (((wo * FO + i) ^ fy) << 5) + sub);
#pragma unroll
  for (int j = 0; j < FC; j++) xa[j] = (uint32_t)(Y_BYTES + k0 * ROWBX + (((wc * FC + j) ^ fx) << 5) + sub);
  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

  typedef short s16x8 __attribute__((ext_vector_type(8)));
  // bias gradient: in the workgroups of input-channel block 0 and tap group 0, the two waves of column half 0 multiply
  // their gy fragments by a fragment of ones as well, which sums them over the step's 32 pixels (pixels past the end of the
  // range were loaded as zeros)
  const bool do_bias = BIAS && p.bias_part != nullptr && ci0 == 0 && tap0 == 0 && wc == 0;
  f32x4 bacc[BIAS ? FO : 1];
#pragma unroll
  for (int i = 0; i < (BIAS ? FO : 1); i++) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);
  auto compute = [&](int stage) {
    const uint32_t sb = smem_base + (uint32_t)(stage * STAGE);
    s16x4 ylo[FO], yhi[FO], xlo[FC], xhi[FC];
#pragma unroll
    for (int i = 0; i < FO; i++) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(ylo[i]) : "v"(sb + ya[i]) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(yhi[i]) : "v"(sb + ya[i]), "n"(4 * ROWBY) : "memory");
    }
#pragma unroll
    for (int j = 0; j < FC; j++) {
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(xlo[j]) : "v"(sb + xa[j]) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(xhi[j]) : "v"(sb + xa[j]), "n"(4 * ROWBX) : "memory");
    }
    // the wait names every destination so that no use can be scheduled ahead of it
    if constexpr (FO == 8)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ylo[0]), "+v"(yhi[0]), "+v"(ylo[1]), "+v"(yhi[1]), "+v"(ylo[2]), "+v"(yhi[2]), "+v"(ylo[3]), "+v"(yhi[3]),
                     "+v"(ylo[4]), "+v"(yhi[4]), "+v"(ylo[5]), "+v"(yhi[5]), "+v"(ylo[6]), "+v"(yhi[6]), "+v"(ylo[7]), "+v"(yhi[7]),
                     "+v"(xlo[0]), "+v"(xhi[0]), "+v"(xlo[1]), "+v"(xhi[1]), "+v"(xlo[2]), "+v"(xhi[2]), "+v"(xlo[3]), "+v"(xhi[3]));
    else if constexpr (FO == 4)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ylo[0]), "+v"(yhi[0]), "+v"(ylo[1]), "+v"(yhi[1]), "+v"(ylo[2]), "+v"(yhi[2]), "+v"(ylo[3]), "+v"(yhi[3]),
                     "+v"(xlo[0]), "+v"(xhi[0]), "+v"(xlo[1]), "+v"(xhi[1]), "+v"(xlo[2]), "+v"(xhi[2]), "+v"(xlo[3]), "+v"(xhi[3]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ylo[0]), "+v"(yhi[0]), "+v"(ylo[1]), "+v"(yhi[1]),
                     "+v"(xlo[0]), "+v"(xhi[0]), "+v"(xlo[1]), "+v"(xhi[1]), "+v"(xlo[2]), "+v"(xhi[2]), "+v"(xlo[3]), "+v"(xhi[3]));
    bf16x8 a[FO], b[FC];
#pragma unroll
    for (int i = 0; i < FO; i++) {
      const s16x8 v = {ylo[i][0], ylo[i][1], ylo[i][2], ylo[i][3], yhi[i][0], yhi[i][1], yhi[i][2], yhi[i][3]};
      a[i] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int j = 0; j < FC; j++) {
      const s16x8 v = {xlo[j][0], xlo[j][1], xlo[j][2], xlo[j][3], xhi[j][0], xhi[j][1], xhi[j][2], xhi[j][3]};
      b[j] = __builtin_bit_cast(bf16x8, v);
    }
#pragma unroll
    for (int i = 0; i < FO; i++)
#pragma unroll
      for (int j = 0; j < FC; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    if constexpr (BIAS) {
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < FO; i++) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], bacc[i], 0, 0, 0);
      }
    }
  };

  // ---- ring: steps it+1 .. it+STAGES-1 are in flight while step it is multiplied
  const int nsteps = kend > kb0 ? (kend - kb0 + kStepPix - 1) / kStepPix : 0;
  int issued = 0, st_i = 0, st_c = 0, kb = kb0;
  auto next_stage = [](int s) { return s + 1 == STAGES ? 0 : s + 1; };
#pragma unroll
  for (int s = 0; s < STAGES - 1; s++)
    if (issued < nsteps) { issue(st_i, kb); st_i = next_stage(st_i); kb += kStepPix; issued++; }
  int it = 0;
  for (; issued < nsteps; it++, issued++) {   // steady state: STAGES-1 steps in flight at the top
    wait_vm<(STAGES - 2) * LOADS>();
    __builtin_amdgcn_s_barrier();      // step `it` has landed for every wave; the buffer of step it-1 is free
    __builtin_amdgcn_sched_barrier(0);
    issue(st_i, kb);
    st_i = next_stage(st_i);
    kb += kStepPix;
    compute(st_c);
    st_c = next_stage(st_c);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; it < nsteps; it++) {                 // drain
    const int younger = nsteps - it - 1;
    if (STAGES >= 4 && younger >= 2) wait_vm<2 * LOADS>();
    else if (STAGES >= 3 && younger >= 1) wait_vm<LOADS>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(st_c);
    st_c = next_stage(st_c);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- partial tile: rows of the accumulator are input channels, so a lane stores 4 consecutive channels
  const int cw_mask = (1 << p.cw_log2) - 1;
#pragma unroll
  for (int i = 0; i < FO; i++) {
    const int co = co0 + wo * (TO / 2) + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < FC; j++) {
      const int col = wc * (kTC / 2) + j * 16 + (lane >> 4) * 4;
      const int tsub = col >> p.cw_log2;
      const int tap = tap0 + tsub, ci = ci0 + (col & cw_mask);
      if (co < p.Cout && tsub < p.tpt && tap < T && ci < p.Cin) {
        float* dst = p.part + (((size_t)split * T + tap) * p.Cout + co) * p.Cin + ci;
        *reinterpret_cast<f32x4*>(dst) = acc[i][j];
      }
    }
  }
  if constexpr (BIAS) {
    if (do_bias && lane < 16) {
#pragma unroll
      for (int i = 0; i < FO; i++) {
        const int co = co0 + wo * (TO / 2) + i * 16 + lane;
        if (co < p.Cout) p.bias_part[(size_t)split * p.Cout + co] = bacc[i][0];
      }
    }
  }
}

template <int TO, int STAGES, bool BIAS>
static int launch_dma_bias(const WgradParams& p, double flops, hipStream_t stream) {
  constexpr int smem = STAGES * kStepPix * (TO + kTC) * 2;
  auto kern = conv_wgrad_dma_kernel<TO, STAGES, BIAS>;
  if (smem > 64 * 1024 && ensure_dynamic_smem(reinterpret_cast<const void*>(kern), smem) != GC_OK) return GC_HIP_ERROR;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_wgrad_dma_kernel<%d, %d>", TO, STAGES);
  ProfScope ps(name, flops, stream);
  const unsigned nblocks = (unsigned)(p.n_co_tiles * p.n_ci_tiles * p.n_tap_groups * p.splits);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kThreads), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_wgrad_dma launch");
}

template <int TO, int STAGES>
static int launch_dma_one(const WgradParams& p, double flops, hipStream_t stream) {
  return p.bias_part != nullptr ? launch_dma_bias<TO, STAGES, true>(p, flops, stream) : launch_dma_bias<TO, STAGES, false>(p, flops, stream);
}

int launch_wgrad_dma(const WgradParams& p, const WgradLaunch& l, hipStream_t s) {
  if (p.K <= 0 || p.Cout <= 0 || p.Cin <= 0) return GC_OK;
  if (l.to == 256) return launch_dma_one<256, 3>(p, l.flops, s);
  if (l.to == 64) return launch_dma_one<64, 4>(p, l.flops, s);
  return launch_dma_one<128, 4>(p, l.flops, s);
}

}  // namespace gc
