// Fast path of the gather-GEMM convolution (forward and data gradient) for bf16 activations whose channel count
// is a multiple of 8: both operands go global -> LDS directly (LDS-DMA, `buffer_load_dwordx4 ... lds`), no register
// staging, no conversion, no LDS writes by the waves.  Same tile geometry, LDS image (128-byte rows, 16-byte
// chunks XOR-swizzled by (row >> 1) & 7) and MFMA loop as conv_gather_kernel; the swizzle is applied on the
// per-lane SOURCE offset because an LDS-DMA wave-instruction writes 1 KiB linearly (8 rows x 128 bytes).
//
// Addressing is kept off the vector ALU: a lane's byte offset into the activation tensor is fixed for the whole
// kernel, the filter tap and channel step move only the scalar offset of the buffer instruction, and a lane whose
// pixel falls into the zero padding (or past the end of the tile) swaps its offset for one beyond the buffer's
// range, which the hardware answers with zeros.  Both tensors therefore have to be smaller than 2 GiB.
//
// A ring of STAGES LDS buffers keeps STAGES-1 reduction steps in flight: the loads of step k+STAGES-1 are issued
// before the MFMAs of step k, a counted `s_waitcnt vmcnt` retires only the oldest step, and one raw `s_barrier`
// per step both publishes the landed tile and frees the buffer that was just consumed.
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "conv_gather.h"

namespace gc {

template <int TO, int TP, int NW>
struct DmaCfg {
  static constexpr int WO = (TO >= 64) ? 2 : 1;
  static constexpr int WP = NW / WO;
  static constexpr int FO = TO / WO / 16;
  static constexpr int FP = TP / WP / 16;
};

__device__ __forceinline__ int dswz(int row) { return (row >> 1) & 7; }

// Which output channel (relative to the tile) the weight row at LDS row `row` holds.  Rows are permuted inside groups of
// 32 so that the accumulators of two neighbouring 16-row MFMA blocks give a lane 8 CONSECUTIVE channels: MFMA block f,
// accumulator row 4g + e  <->  channel 32 (f >> 1) + 8 g + 4 (f & 1) + e.  A lane (g = lane >> 4) then stores its bf16
// results of blocks (f, f + 1) with one 16-byte instruction.  The fragment reads keep their conflict-free row order;
// only the loader's source address and the epilogue's channel index know about the permutation.
__device__ __forceinline__ int oc_of_row(int row) {
  const int f = (row >> 4) & 1, g = (row >> 2) & 3, e = row & 3;
  return (row & ~31) + 8 * g + 4 * f + e;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // s_waitcnt vmcnt(N) alone (expcnt / lgkmcnt fields at "no wait")
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr uint32_t kOutOfRange = 0x80000000u;   // beyond any buffer this kernel accepts: the load returns zeros

// WS ("wave specialised"): the workgroup has 2 NW waves; waves NW.. only issue the LDS-DMA loads, waves 0..NW-1 only
// read fragments and multiply.  A wave that issues loads is held at every load instruction until the texture-address
// path has taken the previous ones (16 cycles per 1 KiB instruction, all waves of the CU in one queue), and a held
// wave issues no MFMA; with the roles split, that wait no longer sits in the matrix pipe's instruction stream.
// Synchronisation stays one workgroup barrier per step, executed by both roles.
template <typename TOut, int TO, int TP, int STAGES, int NW, bool WS = false>
__global__ __launch_bounds__(NW * 64 * (WS ? 2 : 1)) void conv_gather_dma_kernel(const GatherParams p) {
  using C = DmaCfg<TO, TP, NW>;
  constexpr int W_BYTES = TO * 128, A_BYTES = TP * 128;
  constexpr int STAGE = W_BYTES + A_BYTES;
  constexpr int AJ = TP / 8 / NW;  // activation wave-instructions per wave per step (8 rows each)
  constexpr int WJ = TO / 8 / NW;  // weight wave-instructions per wave per step
  constexpr int LOADS = AJ + WJ;  // LDS-DMA instructions per wave per step

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int lane = t & 63, wave_id = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = !WS || wave_id >= NW;      // issues loads
  const bool consumer = !WS || wave_id < NW;     // multiplies and stores
  const int wave = WS && wave_id >= NW ? wave_id - NW : wave_id;   // index within the role
  const int wo = wave / C::WP, wp_ = wave % C::WP;

  const uint32_t nblocks = (uint32_t)p.n_pix_tiles * (uint32_t)p.n_oc_tiles;
  const uint32_t lid = xcd_remap(blockIdx.x, nblocks);
  const int pix_tile = lid / p.n_oc_tiles;
  const int oc0 = (lid % p.n_oc_tiles) * TO;
  int cls = 0, m0;
  if (p.cls_interleave) {   // equal-sized classes: (tile, class) order, the classes of one image region are neighbours
    cls = pix_tile % p.tt.n_classes;
    m0 = (pix_tile / p.tt.n_classes) * TP;
  } else {
#pragma unroll
    for (int c = 1; c < kMaxClasses; c++)
      if (c < p.tt.n_classes && pix_tile >= p.tt.cls_tile_start[c]) cls = c;
    m0 = (pix_tile - p.tt.cls_tile_start[cls]) * TP;
  }
  const int OHc = p.tt.cls_oh[cls], OWc = p.tt.cls_ow[cls];
  const FastDiv div_ow = p.tt.cls_div_ow[cls], div_oh = p.tt.cls_div_oh[cls];
  const int Mc = p.N * OHc * OWc;
  const int tap_begin = p.tt.cls_tap_start[cls], tap_end = p.tt.cls_tap_start[cls + 1];

  // The activation descriptor starts a_bias bytes before the tensor so that every tap's scalar offset is >= 0.
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(p.in)) - p.a_bias, 0, (int)(p.in_bytes + p.a_bias), 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, (int)p.w_bytes, 0x00020000);

  // ---- this lane's rows: wave-instruction j of this wave covers tile rows (wave*J + j)*8 .. +7, lane>>3 picks the row
  const int sub_row = lane >> 3, pos = lane & 7;
  uint32_t a_off[AJ];          // byte offset of (pixel at tap offset (0,0), source chunk) in the activation tensor
  int a_ih[AJ], a_iw[AJ];      // input coordinates at tap offset (0,0); far negative for rows past the end of the class
  int a_chunk[AJ];             // first channel of the source chunk (for the check against the channel count)
#pragma unroll
  for (int j = 0; j < AJ; j++) {
    const int row = (wave * AJ + j) * 8 + sub_row;
    const int m = m0 + row;
    const bool ok = m < Mc;
    const uint32_t mm = ok ? (uint32_t)m : 0u;
    const uint32_t tmp = fd_div(mm, div_ow), n = fd_div(tmp, div_oh);
    const int ow = (int)(mm - tmp * (uint32_t)OWc), oh = (int)(tmp - n * (uint32_t)OHc);
    a_ih[j] = ok ? oh * p.a : -(1 << 20);
    a_iw[j] = ow * p.a;
    a_chunk[j] = (pos ^ dswz(row)) * 8;        // source chunk (elements) that lands at LDS position `pos` of this row
    a_off[j] = (uint32_t)((((size_t)n * p.IH + oh * p.a) * p.IW + ow * p.a) * p.in_cstride + p.in_coff + a_chunk[j]) * 2u;
  }
  uint32_t w_voff[WJ];
#pragma unroll
  for (int j = 0; j < WJ; j++) {
    const int row = (wave * WJ + j) * 8 + sub_row;
    const int oc = oc0 + oc_of_row(row);
    w_voff[j] = oc < p.OC ? (uint32_t)(((size_t)oc * p.row_pitch + (pos ^ dswz(row)) * 8) * 2) : kOutOfRange;
  }

  // The tap table is read through LDS: an ordinary global load inside the loop would make the compiler drain the
  // in-flight LDS-DMA (vmcnt(0)) before using its result.
  int* tap_lds = reinterpret_cast<int*>(smem + STAGES * STAGE);
  if (t < kMaxTaps) tap_lds[t] = ((int)p.tt.dh[t] & 0xff) | (((int)p.tt.dw[t] & 0xff) << 8) | ((int)p.tt.wtap[t] << 16);
  __syncthreads();

  // ---- issue side: position (tap, cs) of the next step to load, with the tap's scalars decoded once per tap
  const int csteps = p.ICp / kBK;
  const int nsteps = (tap_end - tap_begin) * csteps;
  int tap = tap_begin, cs = 0;
  int s_dh = 0, s_dw = 0;
  uint32_t s_aoff = 0, s_woff = 0;
  auto load_tap = [&](int tp) {
    const int e = __builtin_amdgcn_readfirstlane(tap_lds[tp]);
    s_dh = (int)(int8_t)(e & 0xff);
    s_dw = (int)(int8_t)((e >> 8) & 0xff);
    s_aoff = (uint32_t)(((s_dh * p.IW + s_dw) * p.in_cstride) * 2 + p.a_bias);
    s_woff = (uint32_t)((e >> 16) * p.ICp * 2);
  };
  auto issue = [&](int stage) {
    char* wb = smem + stage * STAGE;
    char* ab = wb + W_BYTES;
    const int c0 = cs * kBK;
#pragma unroll
    for (int j = 0; j < WJ; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(wb + (wave * WJ + j) * 1024), 16, (int)w_voff[j],
                                               (int)(s_woff + c0 * 2), 0, 0);
#pragma unroll
    for (int j = 0; j < AJ; j++) {
      // (bitwise on purpose: a short-circuit would turn into exec-mask branches)
      const bool ok = ((unsigned)(a_ih[j] + s_dh) < (unsigned)p.IH) & ((unsigned)(a_iw[j] + s_dw) < (unsigned)p.IW) &
                      ((c0 | a_chunk[j]) < p.IC);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(ab + (wave * AJ + j) * 1024), 16,
                                               (int)(ok ? a_off[j] : kOutOfRange), (int)(s_aoff + c0 * 2), 0, 0);
    }
    if (++cs == csteps) {
      cs = 0;
      if (++tap < tap_end) load_tap(tap);
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
  int issued = 0, st_i = 0, st_c = 0;
  auto next_stage = [](int s) { return s + 1 == STAGES ? 0 : s + 1; };
  if (WS) {
    // Both roles execute one barrier per step: at barrier `it` the loaders have seen step `it` land and the consumers
    // have finished step it-1, whose buffer the loaders then refill.
    if (loader) {
      if (nsteps > 0) load_tap(tap_begin);
#pragma unroll
      for (int s = 0; s < STAGES - 1; s++)
        if (issued < nsteps) { issue(st_i); st_i = next_stage(st_i); issued++; }
      int it = 0;
      for (; issued < nsteps; it++, issued++) {
        wait_vmcnt<(STAGES - 2) * LOADS>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(st_i);
        st_i = next_stage(st_i);
      }
      for (; it < nsteps; it++) {
        const int younger = nsteps - it - 1;
        if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * LOADS>();
        else if (STAGES >= 3 && younger >= 1) wait_vmcnt<LOADS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    for (int it = 0; it < nsteps; it++) {
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      compute(st_c);
      st_c = next_stage(st_c);
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  if (nsteps > 0) load_tap(tap_begin);
#pragma unroll
  for (int s = 0; s < STAGES - 1; s++)
    if (issued < nsteps) { issue(st_i); st_i = next_stage(st_i); issued++; }
  int it = 0;
  // steady state: STAGES-1 steps are in flight at the top, the oldest of them is step `it`
  for (; issued < nsteps; it++, issued++) {
    wait_vmcnt<(STAGES - 2) * LOADS>();
    __builtin_amdgcn_s_barrier();                 // every wave's part of step `it` has landed; step it-1's buffer is free
    __builtin_amdgcn_sched_barrier(0);
    issue(st_i);               // goes into the buffer consumed at step it-1
    st_i = next_stage(st_i);
    compute(st_c);
    st_c = next_stage(st_c);
    __builtin_amdgcn_sched_barrier(0);
  }
  // drain: nothing left to issue
  for (; it < nsteps; it++) {
    const int younger = nsteps - it - 1;
    if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * LOADS>();
    else if (STAGES >= 3 && younger >= 1) wait_vmcnt<LOADS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(st_c);
    st_c = next_stage(st_c);
    __builtin_amdgcn_sched_barrier(0);
  }
  }
  (void)consumer;

  // ---- epilogue ----------------------------------------------------------------------------------------
  // A wave alone on its SIMD issues one vector instruction every 4-5 cycles, so this part is written for instruction
  // count: full tiles with aligned rows (the common case) take a branch-free path, two 16-row blocks per 16-byte store.
  const int g = lane >> 4;
  const bool stats = p.ch_sum != nullptr || p.stat_rows != nullptr;
  TOut* out = reinterpret_cast<TOut*>(p.out);
  const int oc_w = oc0 + wo * (TO / C::WO);          // first channel of this wave's rows
  const bool linear = p.tt.n_classes == 1 && p.b == 1;   // forward pass: output pixel index == GEMM column index
  const bool fast = (C::FO % 2 == 0) && (m0 + TP <= Mc) && (oc0 + TO <= p.OC) && p.out_vec8 != 0 && p.act <= GC_ACT_LRELU;
  const bool scaled = p.pre_scale != nullptr || p.post_scale != nullptr;
  float ssum[C::FO][4], ssq[C::FO][4];
#pragma unroll
  for (int i = 0; i < C::FO; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) ssum[i][r] = ssq[i][r] = 0.f;
  if (fast) {
    // none / relu / leaky relu as one formula: v > 0 ? v : v * s with s = 1, 0, slope
    const float sl = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
    f32x4 bias_v[C::FO];
#pragma unroll
    for (int i = 0; i < C::FO; i++) {
      const int oc = oc_w + (i >> 1) * 32 + g * 8 + (i & 1) * 4;
      bias_v[i] = p.bias != nullptr ? *reinterpret_cast<const f32x4*>(p.bias + oc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int j = 0; j < C::FP; j++) {
      const uint32_t m = (uint32_t)(m0 + wp_ * (TP / C::WP) + j * 16 + (lane & 15));
      size_t pix = m;
      uint32_t n_of = 0;
      if (!linear || scaled) {
        const uint32_t tmp = fd_div(m, div_ow), n = fd_div(tmp, div_oh);
        const int ow = (int)(m - tmp * (uint32_t)OWc), oh = (int)(tmp - n * (uint32_t)OHc);
        pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
        n_of = n;
      }
      const size_t eoff = pix * p.out_cstride + p.out_coff + oc_w + g * 8;
      const size_t soff = (size_t)n_of * p.OC + oc_w + g * 8;     // per-sample scales: [N][OC]
#pragma unroll
      for (int i = 0; i < C::FO; i += 2) {
        f32x4 a = acc[i][j], b = acc[i + 1][j];
        if (p.pre_scale != nullptr) {
          a *= *reinterpret_cast<const f32x4*>(p.pre_scale + soff + (i >> 1) * 32);
          b *= *reinterpret_cast<const f32x4*>(p.pre_scale + soff + (i >> 1) * 32 + 4);
        }
        a += bias_v[i]; b += bias_v[i + 1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          a[r] = a[r] > 0.f ? a[r] : a[r] * sl;
          b[r] = b[r] > 0.f ? b[r] : b[r] * sl;
        }
        if (p.post_scale != nullptr) {
          a *= *reinterpret_cast<const f32x4*>(p.post_scale + soff + (i >> 1) * 32);
          b *= *reinterpret_cast<const f32x4*>(p.post_scale + soff + (i >> 1) * 32 + 4);
        }
        const u32x2 pa = cvt_bf16x4(a), pb = cvt_bf16x4(b);
        if (sizeof(TOut) == 2) {
          *reinterpret_cast<u32x4*>(out + eoff + (i >> 1) * 32) = u32x4{pa[0], pa[1], pb[0], pb[1]};
          if (stats) { a = widen_bf16x4(pa); b = widen_bf16x4(pb); }     // statistics of the stored values
        } else {
          *reinterpret_cast<f32x4*>(out + eoff + (i >> 1) * 32) = a;
          *reinterpret_cast<f32x4*>(out + eoff + (i >> 1) * 32 + 4) = b;
          if (p.out_bf16 != nullptr) *reinterpret_cast<u32x4*>(p.out_bf16 + eoff + (i >> 1) * 32) = u32x4{pa[0], pa[1], pb[0], pb[1]};
        }
        if (stats) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            ssum[i][r] += a[r]; ssq[i][r] += a[r] * a[r];
            ssum[i + 1][r] += b[r]; ssq[i + 1][r] += b[r] * b[r];
          }
        }
      }
    }
  } else {
    const bool vec_ok = p.out_vec != 0;
#pragma unroll
    for (int j = 0; j < C::FP; j++) {
      const int m = m0 + wp_ * (TP / C::WP) + j * 16 + (lane & 15);
      const bool mok = m < Mc;
      const uint32_t mm = mok ? (uint32_t)m : 0u;
      const uint32_t tmp = fd_div(mm, div_ow), n = fd_div(tmp, div_oh);
      const int ow = (int)(mm - tmp * (uint32_t)OWc), oh = (int)(tmp - n * (uint32_t)OHc);
      const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
      TOut* dst = out + pix * p.out_cstride + p.out_coff;
#pragma unroll
      for (int i = 0; i < C::FO; i++) {
        const int oc = oc0 + oc_of_row(wo * (TO / C::WO) + i * 16 + g * 4);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float x = acc[i][j][r];
          if (p.pre_scale != nullptr && oc + r < p.OC) x *= p.pre_scale[(size_t)n * p.OC + oc + r];
          if (p.bias != nullptr && oc + r < p.OC) x += p.bias[oc + r];
          x = apply_act(x, p.act, p.slope);
          if (p.post_scale != nullptr && oc + r < p.OC) x *= p.post_scale[(size_t)n * p.OC + oc + r];
          if (sizeof(TOut) == 2) x = bf16_bits_to_float(float_to_bf16_bits(x));
          v[r] = x;
          if (stats && mok && oc + r < p.OC) { ssum[i][r] += x; ssq[i][r] += x * x; }
        }
        if (!mok) continue;
        if (vec_ok && oc + 3 < p.OC) {
          if (sizeof(TOut) == 4) {
            *reinterpret_cast<f32x4*>(dst + oc) = f32x4{v[0], v[1], v[2], v[3]};
            if (p.out_bf16 != nullptr)
              *reinterpret_cast<u32x2*>(p.out_bf16 + pix * p.out_cstride + p.out_coff + oc) = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]});
          }
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
        const int oc = oc0 + oc_of_row(wo * (TO / C::WO) + i * 16 + g * 4) + r;
        if ((lane & 15) == 0 && oc < p.OC) {
          if (p.stat_rows != nullptr) {      // this wave's pixels of this tile are one row of the partial table
            const size_t row = (size_t)pix_tile * C::WP + wp_;
            *reinterpret_cast<float2*>(p.stat_rows + (row * (size_t)p.OC + oc) * 2) = float2{s, q};
          } else {
            atomicAdd(p.ch_sum + oc, s);
            if (p.ch_sumsq != nullptr) atomicAdd(p.ch_sumsq + oc, q);
          }
        }
      }
  }
}

template <typename TOut, int TO, int TP, int STAGES, int NW, bool WS = false>
static int launch_dma_one(const GatherParams& p, double flops, hipStream_t stream) {
  constexpr int smem = STAGES * (TO + TP) * 128 + kMaxTaps * 4;
  auto kern = conv_gather_dma_kernel<TOut, TO, TP, STAGES, NW, WS>;
  if (smem > 64 * 1024 && ensure_dynamic_smem(reinterpret_cast<const void*>(kern), smem) != GC_OK) return GC_HIP_ERROR;
  const unsigned nblocks = (unsigned)p.n_pix_tiles * (unsigned)p.n_oc_tiles;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_gather_dma_kernel<%s, %d, %d, %d, %d%s>", sizeof(TOut) == 4 ? "float" : "unsigned short", TO, TP, STAGES,
           NW, WS ? ", true" : "");
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NW * 64 * (WS ? 2 : 1)), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_gather_dma launch");
}

int launch_gather_dma(const GatherParams& p0, const GatherLaunch& l, bool out_f32, hipStream_t s) {
  if (p0.n_pix_tiles == 0 || p0.n_oc_tiles == 0) return GC_OK;
  GatherParams p = p0;
  if (p.stat_rows != nullptr) {     // one row per consumer wave's pixel share of a tile; no rows when they do not fit the caller's table
    const int nw = (l.to == 256 || l.tp == 256) ? 8 : 4;
    const long rows = (long)p.n_pix_tiles * (nw / 2);
    if (rows <= p.stat_cap) *p.stat_written = (int32_t)rows;
    else p.stat_rows = nullptr;
  }
  if (l.to == 256 && l.tp == 256) return out_f32 ? launch_dma_one<float, 256, 256, 2, 8>(p, l.flops, s) : launch_dma_one<uint16_t, 256, 256, 2, 8>(p, l.flops, s);
  if (l.to == 128 && l.tp == 256) return out_f32 ? launch_dma_one<float, 128, 256, 3, 8>(p, l.flops, s) : launch_dma_one<uint16_t, 128, 256, 3, 8>(p, l.flops, s);
  // At most one workgroup per CU: nothing else hides the load latency, so spend the idle LDS on a deeper ring.
  const bool deep = (long)p.n_pix_tiles * p.n_oc_tiles <= 256;
  if (l.to == 128 && deep) return out_f32 ? launch_dma_one<float, 128, 128, 4, 4>(p, l.flops, s) : launch_dma_one<uint16_t, 128, 128, 4, 4>(p, l.flops, s);
  if (l.to == 128) return out_f32 ? launch_dma_one<float, 128, 128, 2, 4>(p, l.flops, s) : launch_dma_one<uint16_t, 128, 128, 2, 4>(p, l.flops, s);
  if (l.to == 64 && deep) return out_f32 ? launch_dma_one<float, 64, 128, 4, 4>(p, l.flops, s) : launch_dma_one<uint16_t, 64, 128, 4, 4>(p, l.flops, s);
  if (l.to == 64) return out_f32 ? launch_dma_one<float, 64, 128, 3, 4, true>(p, l.flops, s) : launch_dma_one<uint16_t, 64, 128, 3, 4, true>(p, l.flops, s);
  set_error("internal: no LDS-DMA variant for this tile");
  return GC_UNSUPPORTED_GEOMETRY;
}

}  // namespace gc
