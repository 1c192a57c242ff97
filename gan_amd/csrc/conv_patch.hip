// Patch-resident gather convolution for stride-1 gathers with several taps per class: 3x3 stride-1 forward / data
// gradient, and the 2x2-tap parity classes of a 4x4 stride-2 data gradient (= transposed-conv forward).
//
// The tiled LDS-DMA kernel (conv_gather_dma.hip) re-loads a tile's pixels once per tap: a 3x3 layer moves every input
// element into LDS nine times, and on large layers that traffic (L2 -> LDS at about 40 GB/s per CU chip-wide), not the
// matrix cores, sets the pace.  Here a workgroup owns an 8 x 32 rectangle of one sample's (class-local) output and TO
// (128 or 64) output channels.  Per 32-channel chunk it loads the rectangle's input PATCH once ((8 + dh range) x (32 + dw
// range) pixels, zeros outside the image) and forms the B operand of every tap from it at a shifted LDS address; only
// the weights ([TO oc][32 ch] per tap, 8 / 4 KiB) stream, one tap per step.
//
// Two workgroups of four waves share a CU (76 KiB of LDS, at most 256 registers).  They are independent, so one's fixed
// parts -- argument reads, the first patch, the output stores, the barrier of every step -- run beside the other's
// MFMAs; an earlier form with one workgroup of eight waves on a 16 x 32 rectangle per CU spent 1.5-4x the MFMA time per
// rectangle on layers with few chunks (64 -> 128 at 256 x 256: 506 TFLOP/s).
//
// Pipeline: weights go through a 4-slot ring (three steps in flight), the patch is double-buffered and the next
// chunk's patch is issued at the first step of the current one; the LDS fragment reads of step s + 1 are issued in front
// of the MFMAs of step s (two register sets), so a wave's matrix instructions never wait for its own LDS reads.  All loads are LDS-DMA with buffer addressing (per-lane
// offsets fixed, chunk / tap moves only the scalar offset, masked lanes take an out-of-range offset and receive zeros);
// past the last step the cursor keeps issuing out-of-range loads into free slots so that the counted waits stay uniform.
// LDS rows are 64 bytes (32 channels); the 16-byte chunk position is XORed with (row >> 1) & 3 on the source side: with
// the lane groups a ds_read_b128 is served in, the fragments of 16 consecutive rows starting anywhere are then
// conflict-free (modelled in tools/lds_conflicts.py).  Wave w computes all TO channels for rows 2w, 2w + 1 of the
// rectangle (64 pixels): TO / 4 MFMAs per tap against TO / 16 + 4 fragment reads.
// Weight rows are permuted inside groups of 32 (oc_of_row) so that a lane's two MFMA blocks of a group hold 8
// consecutive channels of a pixel: one 16-byte store per (pixel, 8 channels) in bf16.
//
// S2D: the forward pass of a 4x4 / stride 2 / padding 1 convolution (the down-sampling layers of the critics and of the
// U-Net encoder) as the 2x2-tap stride-1 convolution it is on the space-to-depth image: output pixel (y, x), filter tap
// (2 tr + dy, 2 ts + dx) reads input pixel (2 (y + tr) - 1 + dy, 2 (x + ts) - 1 + dx), i.e. pixel (y + tr, x + ts) of the
// image of 2x2 blocks whose 4 C channels are (dy, dx, c).  Nothing is rearranged in memory: a "chunk" of that image is a
// (dy, dx, 32 channels) triple, its patch (9 x 33 block pixels) is gathered with a pixel stride of two rows / columns, and
// its weights are the taps (2 tr + dy, 2 ts + dx) of the ordinary packed image.  The multiply loop is the same.
#include <stdio.h>

#include "conv_gather.h"

namespace gc {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr uint32_t kOutOfRange = 0x80000000u;
constexpr int kTH = 8, kTW = 32, kCK = 32;                 // rectangle, channels per chunk
constexpr int kNW = 4;                                      // waves
constexpr int kPInstr = 22;                                 // patch DMA instructions of 16 pixels per chunk
constexpr int kMaxPatch = kPInstr * 16;                     // 352 pixels
constexpr int kPatchBytes = kPInstr * 1024;                 // 22 KiB per buffer
constexpr int kWSlots = 4;
constexpr int smem_bytes(int to) { return 2 * kPatchBytes + kWSlots * to * kCK * 2 + kMaxTaps * 4; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// channel (relative to the tile) held by the weight row at LDS row `row`: MFMA block f = row >> 4, accumulator row
// 4 g + e  <->  channel 32 (f >> 1) + 8 g + 4 (f & 1) + e
__device__ __forceinline__ int oc_of_row(int row) {
  const int f = (row >> 4) & 1, g = (row >> 2) & 3, e = row & 3;
  return (row & ~31) + 8 * g + 4 * f + e;
}

template <typename TOut, int TO, bool MASK, bool S2D = false>
__global__ __launch_bounds__(kNW * 64, 2) void conv_patch_kernel(const GatherParams p) {
  constexpr int FO = TO / 16;              // MFMA blocks of output channels
  constexpr int WJ = TO / 64;              // weight DMA instructions per wave and step
  constexpr int kWBytes = TO * kCK * 2;    // one tap
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* patch_base = smem;
  char* w_base = smem + 2 * kPatchBytes;
  int* tap_lds = reinterpret_cast<int*>(w_base + kWSlots * kWBytes);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, c16 = lane & 15;

  if (t < kMaxTaps) tap_lds[t] = ((int)p.tt.dh[t] & 0xff) | (((int)p.tt.dw[t] & 0xff) << 8) | ((int)p.tt.wtap[t] << 16);
  __syncthreads();

  // ---- which (class, sample, rectangle, channel tile)
  const uint32_t nblocks = (uint32_t)p.n_pix_tiles * (uint32_t)p.n_oc_tiles;
  const uint32_t lid = xcd_remap(blockIdx.x, nblocks);
  const int pix_tile = lid / p.n_oc_tiles;
  const int oc0 = (lid % p.n_oc_tiles) * TO;
  int cls = 0;
#pragma unroll
  for (int c = 1; c < kMaxClasses; c++)
    if (c < p.tt.n_classes && pix_tile >= p.tt.cls_tile_start[c]) cls = c;
  const int OHc = p.tt.cls_oh[cls], OWc = p.tt.cls_ow[cls];
  const int tiles_w = (OWc + kTW - 1) / kTW, tiles_h = (OHc + kTH - 1) / kTH;
  int rest = pix_tile - p.tt.cls_tile_start[cls];
  const int tw_ = rest % tiles_w; rest /= tiles_w;
  const int th_ = rest % tiles_h;
  const int n = rest / tiles_h;
  const int tap_begin = p.tt.cls_tap_start[cls], tap_end = p.tt.cls_tap_start[cls + 1];
  int T = tap_end - tap_begin;
  int dh_min = 127, dh_max = -128, dw_min = 127, dw_max = -128;
  if (S2D) {
    T = 4; dh_min = dw_min = 0; dh_max = dw_max = 1;       // 2 x 2 block taps
  } else {
    for (int e = tap_begin; e < tap_end; e++) {
      const int v = __builtin_amdgcn_readfirstlane(tap_lds[e]);
      const int dh = (int)(int8_t)(v & 0xff), dw = (int)(int8_t)((v >> 8) & 0xff);
      dh_min = min(dh_min, dh); dh_max = max(dh_max, dh);
      dw_min = min(dw_min, dw); dw_max = max(dw_max, dw);
    }
  }
  const int PH = kTH + dh_max - dh_min, PW = kTW + dw_max - dw_min;
  const int npatch = PH * PW;
  // The taps of a class form a grid in row-major order (checked by plan_patch): tap (r, s) = r * nw + s has the patch
  // offset to0 + r * dto_r + s * dto_s and the filter tap wt0 + r * dwt_r + s * dwt_s.  Kept in scalar registers, so no
  // step waits for a table read.
  const int nw = dw_max - dw_min + 1;
  int to0, dto_r = 0, dto_s = 0, wt0, dwt_r = 0, dwt_s = 0;
  if (S2D) {
    to0 = 0; dto_r = PW; dto_s = 1;
    wt0 = 0; dwt_r = 8; dwt_s = 2;                         // filter tap (2 tr + dy) * 4 + 2 ts + dx; (dy, dx) comes with the chunk
  } else {
    auto tap_at = [&](int i, int& to, int& wt) {
      const int v = __builtin_amdgcn_readfirstlane(tap_lds[tap_begin + i]);
      to = ((int)(int8_t)(v & 0xff) - dh_min) * PW + ((int)(int8_t)((v >> 8) & 0xff) - dw_min);
      wt = v >> 16;
    };
    tap_at(0, to0, wt0);
    int t1, w1;
    if (nw > 1) { tap_at(1, t1, w1); dto_s = t1 - to0; dwt_s = w1 - wt0; }
    if (T > nw) { tap_at(nw, t1, w1); dto_r = t1 - to0; dwt_r = w1 - wt0; }
  }

  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, (int)p.w_bytes, 0x00020000);

  // ---- per-lane source offsets: patch instruction j * 4 + wave covers patch pixels 16 (j * 4 + wave) .. + 15
  // (waves 0 and 1 issue six, waves 2 and 3 five of the 22 instructions)
  const int npi = wave < 2 ? 6 : 5;
  uint32_t a_off[6];
  int a_chunk[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int px = 16 * (j * kNW + wave) + (lane >> 2);
    const int pos = lane & 3;
    const int chunk = pos ^ ((px >> 1) & 3);
    const int pr = px / PW, pc = px - pr * PW;
    if (S2D) {
      // block pixel (by, bx): the offset is that of its sub-pixel (dy, dx) = (1, 1), input pixel (2 by, 2 bx) (non-negative;
      // the other three lie one row / column in front of it); bits 3 .. 6 say which of the four lie inside the image
      const int ih = 2 * (th_ * kTH + pr), iw = 2 * (tw_ * kTW + pc);
      int okb = 0;
#pragma unroll
      for (int d = 0; d < 4; d++)
        okb |= ((px < npatch) & ((unsigned)(ih - 1 + (d >> 1)) < (unsigned)p.IH) & ((unsigned)(iw - 1 + (d & 1)) < (unsigned)p.IW)) ? (8 << d) : 0;
      a_chunk[j] = chunk * 8 + (okb << 4);               // bits 0 .. 6: channel offset of the 16-byte piece, bits 7 .. 10: validity
      a_off[j] = (uint32_t)((((size_t)n * p.IH + ih) * p.IW + iw) * p.in_cstride + p.in_coff + chunk * 8) * 2u;
    } else {
      const int ih = th_ * kTH + pr + dh_min, iw = tw_ * kTW + pc + dw_min;
      const bool ok = (px < npatch) & ((unsigned)ih < (unsigned)p.IH) & ((unsigned)iw < (unsigned)p.IW);
      a_chunk[j] = ok ? chunk * 8 : (1 << 30);     // a masked lane fails the channel check of every chunk
      a_off[j] = (uint32_t)((((size_t)n * p.IH + (ok ? ih : 0)) * p.IW + (ok ? iw : 0)) * p.in_cstride + p.in_coff + chunk * 8) * 2u;
    }
  }
  uint32_t w_voff[WJ];
#pragma unroll
  for (int j = 0; j < WJ; j++) {
    const int row = (wave * WJ + j) * 16 + (lane >> 2), pos = lane & 3;
    const int oc = oc0 + oc_of_row(row);
    w_voff[j] = oc < p.OC ? (uint32_t)(((size_t)oc * p.row_pitch + (pos ^ ((row >> 1) & 3)) * 8) * 2) : kOutOfRange;
  }

  const int nch = p.ICp / kCK;                             // 32-channel chunks of the input
  const int nchunks = S2D ? 4 * nch : nch;                 // S2D: chunk = (dy, dx, 32 channels), the channel chunk running fastest
  const int nsteps = nchunks * T;

  auto issue_patch = [&](int c) {
    char* pb = patch_base + (c & 1) * kPatchBytes;
    if (S2D) {
      const int d = c / nch, c0 = (c - d * nch) * kCK;
      // sub-pixel (dy, dx) lies (1 - dy) rows and (1 - dx) columns in front of the lane's anchor
      const uint32_t back = (uint32_t)(((1 - (d >> 1)) * p.IW + (1 - (d & 1))) * p.in_cstride) * 2u;
#pragma unroll
      for (int j = 0; j < 6; j++)
        if (j < 5 || npi == 6) {
          const bool ok = ((a_chunk[j] >> (7 + d)) & 1) && (c0 + (a_chunk[j] & 127)) < p.IC;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(pb + (j * kNW + wave) * 1024), 16,
                                                   (int)(ok ? a_off[j] - back : kOutOfRange), c0 * 2, 0, 0);
        }
      return;
    }
    const int c0 = c * kCK;
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j < 5 || npi == 6)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(pb + (j * kNW + wave) * 1024), 16,
                                                 (int)((c0 + a_chunk[j]) < p.IC ? a_off[j] : kOutOfRange), c0 * 2, 0, 0);
  };
  // weights of step (chunk c, tap ti): [TO oc][32 ch]; steps past the last one load zeros (uniform wait counts)
  auto issue_w = [&](int slot, int c, int tr, int ts, bool live) {
    uint32_t soff;
    if (S2D) {
      const int d = c / nch, c0 = (c - d * nch) * kCK;
      soff = (uint32_t)((((d >> 1) * 4 + (d & 1) + tr * dwt_r + ts * dwt_s) * p.ICp + c0) * 2);
    } else {
      soff = (uint32_t)(((wt0 + tr * dwt_r + ts * dwt_s) * p.ICp + c * kCK) * 2);
    }
#pragma unroll
    for (int j = 0; j < WJ; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(w_base + slot * kWBytes + (wave * WJ + j) * 1024), 16,
                                               (int)(live ? w_voff[j] : kOutOfRange), (int)(live ? soff : 0u), 0, 0);
  };

  f32x4 acc[FO][4];
#pragma unroll
  for (int i = 0; i < FO; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // patch index of this lane's pixel in each of its four 16-pixel groups (tap offset (dh_min, dw_min))
  int base_idx[4];
#pragma unroll
  for (int pg = 0; pg < 4; pg++) base_idx[pg] = (2 * wave + (pg >> 1)) * PW + (pg & 1) * 16 + c16;
  // weight fragment offsets inside a tap's [TO][32] tile
  int a_addr[FO];
#pragma unroll
  for (int ob = 0; ob < FO; ob++) {
    const int row = ob * 16 + c16;
    a_addr[ob] = row * 64 + ((g ^ ((row >> 1) & 3)) << 4);
  }

  // fragments of step (chunk c, tap ti) out of ring slot `slot`: TO / 16 weight blocks and the four pixel groups
  auto load_frags = [&](int slot, int c, int tr, int ts, bf16x8 (&a)[FO], bf16x8 (&b)[4]) {
    const char* pb = patch_base + (c & 1) * kPatchBytes;
    const char* wb = w_base + slot * kWBytes;
    const int toff = to0 + tr * dto_r + ts * dto_s;
#pragma unroll
    for (int pg = 0; pg < 4; pg++) {
      const int idx = base_idx[pg] + toff;
      b[pg] = *reinterpret_cast<const bf16x8*>(pb + idx * 64 + ((g ^ ((idx >> 1) & 3)) << 4));
    }
#pragma unroll
    for (int ob = 0; ob < FO; ob++) a[ob] = *reinterpret_cast<const bf16x8*>(wb + a_addr[ob]);
  };
  auto mfmas = [&](const bf16x8 (&a)[FO], const bf16x8 (&b)[4]) {
#pragma unroll
    for (int ob = 0; ob < FO; ob++)
#pragma unroll
      for (int pg = 0; pg < 4; pg++) acc[ob][pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ob], b[pg], acc[ob][pg], 0, 0, 0);
  };

  // ---- prologue: patch 0, the weights of steps 0 .. 3, the fragments of step 0
  issue_patch(0);
  int ic = 0, itr = 0, its = 0, islot = 0, issued = 0;     // (chunk, tap row, tap column, ring slot) of the next weight step to issue
  auto issue_next_w = [&]() {
    issue_w(islot, ic, itr, its, issued < nsteps);
    issued++;
    islot = (islot + 1) & (kWSlots - 1);
    if (++its == nw) {
      its = 0;
      if (++itr * nw == T) { itr = 0; ic++; }
    }
  };
#pragma unroll
  for (int s = 0; s < kWSlots; s++) issue_next_w();
  bf16x8 a0[FO], b0[4], a1[FO], b1[4];
  wait_vmcnt<3 * WJ>();                  // the patch and the weights of step 0
  __builtin_amdgcn_s_barrier();
  load_frags(0, 0, 0, 0, a0, b0);
  // ---- steps.  The fragments of step s + 1 are read while the MFMAs of step s run, so at the barrier of step s only the
  // slot of step s + 1 is still needed: the weights of s + 2 and s + 3 are in flight and those of s + 4 go into the slot
  // of step s (whose fragments sit in registers since the previous step).
  int cc = 0, cti = 0;                   // (chunk, tap) of the step being multiplied
  int nc = 0, ntr = 0, nts = 0, nslot = 1;   // (chunk, tap row, tap column, ring slot) of the step whose fragments are read next
  auto next_frag_tap = [&]() {
    if (++nts == nw) {
      nts = 0;
      if (++ntr * nw == T) { ntr = 0; nc++; }
    }
  };
  next_frag_tap();
  auto step = [&](int s, const bf16x8 (&a)[FO], const bf16x8 (&b)[4], bf16x8 (&an)[FO], bf16x8 (&bn)[4]) {
    // Younger than the weights of step s + 1: those of steps s + 2 and s + 3, and the next chunk's patch when it went out at
    // one of the two previous steps (this is tap 1 or 2 of a chunk that has a successor).
    if ((cti == 1 || cti == 2) && cc + 1 < nchunks) {
      if (npi == 6) wait_vmcnt<2 * WJ + 6>(); else wait_vmcnt<2 * WJ + 5>();
    } else {
      wait_vmcnt<2 * WJ>();
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's reads of the slot about to be refilled have returned
    __builtin_amdgcn_s_barrier();        // step s + 1 has landed for every wave; the slot of step s and the other patch buffer are free
    __builtin_amdgcn_sched_barrier(0);
    if (cti == 0 && cc + 1 < nchunks) issue_patch(cc + 1);
    issue_next_w();
    if (s + 1 < nsteps) load_frags(nslot, nc, ntr, nts, an, bn);
    mfmas(a, b);
    nslot = (nslot + 1) & (kWSlots - 1);
    next_frag_tap();
    if (++cti == T) { cti = 0; cc++; }
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int s = 0; s < nsteps; s += 2) {
    step(s, a0, b0, a1, b1);
    if (s + 1 < nsteps) step(s + 1, a1, b1, a0, b0);
  }
  wait_vmcnt<0>();                       // the zero-filling tail loads target this workgroup's LDS: let them land

  // ---- statistics rows (a BatchNorm behind this convolution): per 32-channel group the four pixel groups in turn, so that
  // only 16 sums are live beside the accumulators; a row of the table is this wave's 64 pixels, summed over the 16 lanes
  // that share a channel.  The planner grants rows only with 16-byte stores, whole rectangles and no mask.
  if (!MASK && p.stat_rows != nullptr) {
    TOut* out = reinterpret_cast<TOut*>(p.out);
    const float ls = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
    const bool lean = p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || p.act == GC_ACT_LRELU;
    const size_t row = (size_t)pix_tile * kNW + wave;
#pragma unroll
    for (int m = 0; m < FO / 2; m++) {
      const int oc = oc0 + 32 * m + 8 * g;
      if (oc + 7 >= p.OC) continue;        // (OC is a multiple of 8 here: the lane's eight channels lie inside or outside)
      f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
      if (p.bias != nullptr) { b0 = *reinterpret_cast<const f32x4*>(p.bias + oc); b1 = *reinterpret_cast<const f32x4*>(p.bias + oc + 4); }
      float ssum[8], ssq[8];
#pragma unroll
      for (int r = 0; r < 8; r++) ssum[r] = ssq[r] = 0.f;
#pragma unroll
      for (int pg = 0; pg < 4; pg++) {
        const int oh = th_ * kTH + 2 * wave + (pg >> 1), ow = tw_ * kTW + (pg & 1) * 16 + c16;
        const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
        TOut* dst = out + pix * p.out_cstride + p.out_coff;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) { v[r] = acc[2 * m][pg][r] + b0[r]; v[4 + r] = acc[2 * m + 1][pg][r] + b1[r]; }
        if (lean) {
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], v[r] * ls);
        } else {
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] = apply_act(v[r], p.act, p.slope);
        }
        if (sizeof(TOut) == 4) {
          *reinterpret_cast<f32x4*>(dst + oc) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(dst + oc + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
          *reinterpret_cast<u32x4*>(dst + oc) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          const f32x4 w0 = widen_bf16x4(lo), w1 = widen_bf16x4(hi);      // statistics of the stored values
#pragma unroll
          for (int r = 0; r < 4; r++) { v[r] = w0[r]; v[4 + r] = w1[r]; }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) { ssum[r] += v[r]; ssq[r] += v[r] * v[r]; }
      }
#pragma unroll
      for (int r = 0; r < 8; r++) {
        float sm = ssum[r], sq = ssq[r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          sm += __shfl_xor(sm, d, 64);
          sq += __shfl_xor(sq, d, 64);
        }
        if (c16 == 0) *reinterpret_cast<float2*>(p.stat_rows + (row * (size_t)p.OC + oc + r) * 2) = float2{sm, sq};
      }
    }
    return;
  }

  // ---- epilogue: per (pixel group, group of 32 channels) a lane owns 8 consecutive channels of one pixel
  TOut* out = reinterpret_cast<TOut*>(p.out);
  const bool vec8 = p.out_vec8 != 0;
  const bool lean = p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || p.act == GC_ACT_LRELU;
  const float ls = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
#pragma unroll
  for (int pg = 0; pg < 4; pg++) {
    const int oh = th_ * kTH + 2 * wave + (pg >> 1), ow = tw_ * kTW + (pg & 1) * 16 + c16;
    if (oh >= OHc || ow >= OWc) continue;
    const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
    TOut* dst = out + pix * p.out_cstride + p.out_coff;
#pragma unroll
    for (int m = 0; m < FO / 2; m++) {
      const int oc = oc0 + 32 * m + 8 * g;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; r++) { v[r] = acc[2 * m][pg][r]; v[4 + r] = acc[2 * m + 1][pg][r]; }
      if (vec8 && oc + 7 < p.OC) {
        if (p.pre_scale != nullptr) {
          const float* sc = p.pre_scale + (size_t)n * p.OC + oc;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
#pragma unroll
          for (int r = 0; r < 4; r++) { v[r] *= s0[r]; v[4 + r] *= s1[r]; }
        }
        if (p.bias != nullptr) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + oc), b1 = *reinterpret_cast<const f32x4*>(p.bias + oc + 4);
#pragma unroll
          for (int r = 0; r < 4; r++) { v[r] += b0[r]; v[4 + r] += b1[r]; }
        }
        if (lean) {
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], v[r] * ls);
        } else {
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] = apply_act(v[r], p.act, p.slope);
        }
        if (p.post_scale != nullptr) {
          const float* sc = p.post_scale + (size_t)n * p.OC + oc;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
#pragma unroll
          for (int r = 0; r < 4; r++) { v[r] *= s0[r]; v[4 + r] *= s1[r]; }
        }
        if (MASK) {
          const TOut* mk = reinterpret_cast<const TOut*>(p.mask) + pix * p.out_cstride + p.out_coff + oc;
          float mv[8];
          if (sizeof(TOut) == 4) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(mk), m1 = *reinterpret_cast<const f32x4*>(mk + 4);
#pragma unroll
            for (int r = 0; r < 4; r++) { mv[r] = m0[r]; mv[4 + r] = m1[r]; }
          } else {
            const u32x4 q = *reinterpret_cast<const u32x4*>(mk);
#pragma unroll
            for (int r = 0; r < 4; r++) {
              mv[2 * r] = __builtin_bit_cast(float, q[r] << 16);
              mv[2 * r + 1] = __builtin_bit_cast(float, q[r] & 0xffff0000u);
            }
          }
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] = mv[r] > 0.f ? v[r] : v[r] * p.mask_slope;
        }
        if (sizeof(TOut) == 4) {
          *reinterpret_cast<f32x4*>(dst + oc) = f32x4{v[0], v[1], v[2], v[3]};
          *reinterpret_cast<f32x4*>(dst + oc + 4) = f32x4{v[4], v[5], v[6], v[7]};
        } else {
          const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
          *reinterpret_cast<u32x4*>(dst + oc) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      } else {
#pragma unroll
        for (int r = 0; r < 8; r++)
          if (oc + r < p.OC) {
            float x = v[r];
            if (p.pre_scale != nullptr) x *= p.pre_scale[(size_t)n * p.OC + oc + r];
            if (p.bias != nullptr) x += p.bias[oc + r];
            x = apply_act(x, p.act, p.slope);
            if (p.post_scale != nullptr) x *= p.post_scale[(size_t)n * p.OC + oc + r];
            if (sizeof(TOut) == 4) reinterpret_cast<float*>(dst)[oc + r] = x;
            else reinterpret_cast<uint16_t*>(dst)[oc + r] = float_to_bf16_bits(x);
          }
      }
    }
  }
}

template <typename TOut, int TO, bool MASK, bool S2D = false>
int launch_one(const GatherParams& p, double flops, hipStream_t stream) {
  constexpr int kSmem = smem_bytes(TO);
  auto kern = conv_patch_kernel<TOut, TO, MASK, S2D>;
  if (ensure_dynamic_smem(reinterpret_cast<const void*>(kern), kSmem) != GC_OK) return GC_HIP_ERROR;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_patch_kernel<%s, %d, %s%s>", sizeof(TOut) == 4 ? "float" : "unsigned short", TO, MASK ? "true" : "false",
           S2D ? ", strided" : "");
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)p.n_pix_tiles * (unsigned)p.n_oc_tiles), dim3(kNW * 64), kSmem, stream, p);
  return check_hip(hipGetLastError(), "conv_patch launch");
}

template <typename TOut, int TO>
int launch_mask(const GatherParams& p, double flops, bool mask, hipStream_t stream) {
  return mask ? launch_one<TOut, TO, true>(p, flops, stream) : launch_one<TOut, TO, false>(p, flops, stream);
}

int patch_to(const GatherParams& p) {
  static const bool no64 = getenv("GC_PATCH_NO64") != nullptr;
  if (p.OC >= 128) return 128;
  return (p.OC >= 64 && !no64) ? 64 : 0;
}

}  // namespace

// Decides whether the patch kernel takes the problem and, if so, replaces the tile bookkeeping in p (class tile
// prefix, tile counts) by its own.  The caller has established: bf16 input, 16-byte aligned channel slices, both
// tensors below 2 GiB, no reduction split.
bool plan_patch(GatherParams& p, const GatherLaunch& l) {
  const int to = patch_to(p);
  if (!l.in_bf16 || l.mode != kVecLoads || (p.a != 1 && p.a != 2) || to == 0 || p.ICp % kCK != 0) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.out_bf16 != nullptr) return false;
  long tiles = 0;
  int starts[kMaxClasses + 1];
  if (p.a == 2) {
    // the strided mode: a 4x4 / stride 2 / padding 1 forward convolution with its taps in filter order
    static const bool off = getenv("GC_PATCH_NO_STRIDED") != nullptr;
    if (off || p.b != 1 || p.tt.n_classes != 1 || p.tt.cls_tap_start[1] - p.tt.cls_tap_start[0] != 16) return false;
    const int t0 = p.tt.cls_tap_start[0];
    for (int e = 0; e < 16; e++)
      if (p.tt.dh[t0 + e] != e / 4 - 1 || p.tt.dw[t0 + e] != e % 4 - 1 || p.tt.wtap[t0 + e] != e) return false;
    const int oh = p.tt.cls_oh[0], ow = p.tt.cls_ow[0];
    if (oh % kTH != 0 || ow % kTW != 0 || p.tt.cls_ph[0] != 0 || p.tt.cls_pw[0] != 0) return false;
    // a lane's anchor pixel (2 by, 2 bx) may lie one row and one column past the image: its 32-bit offset must not wrap
    if (p.in_bytes + ((int64_t)p.IW + 2) * p.in_cstride * 2 >= (1ll << 31)) return false;
    tiles = (long)p.N * (oh / kTH) * (ow / kTW);
    const int n_oc_tiles = (p.OC + to - 1) / to;
    if (tiles * n_oc_tiles < 512 || tiles * n_oc_tiles >= (1l << 31)) return false;
    p.tt.cls_tile_start[0] = 0;
    for (int c = 1; c <= kMaxClasses; c++) p.tt.cls_tile_start[c] = (int)tiles;
    p.n_pix_tiles = (int)tiles;
    p.n_oc_tiles = n_oc_tiles;
    p.cls_interleave = 0;
    return true;
  }
  for (int c = 0; c < p.tt.n_classes; c++) {
    const int t0 = p.tt.cls_tap_start[c], t1 = p.tt.cls_tap_start[c + 1];
    if (t1 - t0 < 4) return false;       // the wait counts assume that a chunk's patch has three steps to land
    int dh_min = 127, dh_max = -128, dw_min = 127, dw_max = -128;
    for (int e = t0; e < t1; e++) {
      dh_min = p.tt.dh[e] < dh_min ? p.tt.dh[e] : dh_min; dh_max = p.tt.dh[e] > dh_max ? p.tt.dh[e] : dh_max;
      dw_min = p.tt.dw[e] < dw_min ? p.tt.dw[e] : dw_min; dw_max = p.tt.dw[e] > dw_max ? p.tt.dw[e] : dw_max;
    }
    if ((kTH + dh_max - dh_min) * (kTW + dw_max - dw_min) > kMaxPatch) return false;
    {   // taps in row-major grid order, offsets and filter taps affine in (row, column): the kernel derives them from three entries
      const int nw = dw_max - dw_min + 1, nh = dh_max - dh_min + 1;
      if (nw * nh != t1 - t0) return false;
      const int ddh_r = nh > 1 ? p.tt.dh[t0 + nw] - p.tt.dh[t0] : 0, ddw_s = nw > 1 ? p.tt.dw[t0 + 1] - p.tt.dw[t0] : 0;
      const int dwt_r = nh > 1 ? (int)p.tt.wtap[t0 + nw] - (int)p.tt.wtap[t0] : 0, dwt_s = nw > 1 ? (int)p.tt.wtap[t0 + 1] - (int)p.tt.wtap[t0] : 0;
      for (int i = 0; i < t1 - t0; i++) {
        const int r = i / nw, q = i % nw;
        if (p.tt.dh[t0 + i] != p.tt.dh[t0] + r * ddh_r || p.tt.dw[t0 + i] != p.tt.dw[t0] + q * ddw_s ||
            (int)p.tt.wtap[t0 + i] != (int)p.tt.wtap[t0] + r * dwt_r + q * dwt_s)
          return false;
      }
    }
    const int oh = p.tt.cls_oh[c], ow = p.tt.cls_ow[c];
    if (oh % kTH != 0 || ow % kTW != 0) return false;    // partial rectangles would idle most of a workgroup
    starts[c] = (int)tiles;
    tiles += (long)p.N * (oh / kTH) * (ow / kTW);
  }
  const int n_oc_tiles = (p.OC + to - 1) / to;
  if (tiles * n_oc_tiles < 512 || tiles * n_oc_tiles >= (1l << 31)) return false;   // fewer: the tiled kernels fill the chip better
  for (int c = 0; c < p.tt.n_classes; c++) p.tt.cls_tile_start[c] = starts[c];
  for (int c = p.tt.n_classes; c <= kMaxClasses; c++) p.tt.cls_tile_start[c] = (int)tiles;
  p.n_pix_tiles = (int)tiles;
  p.n_oc_tiles = n_oc_tiles;
  p.cls_interleave = 0;
  return true;
}

// Applies p.mask itself (and clears it) when the 16-byte store path holds for every lane; otherwise the caller's sweep does.
int launch_patch(GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  if (p.n_pix_tiles == 0 || p.n_oc_tiles == 0) return GC_OK;
  if (p.stat_rows != nullptr) {          // one row per wave (64 pixels of a rectangle); no rows when the table is too small
    const long rows = (long)p.n_pix_tiles * kNW;
    const bool ok = p.mask == nullptr && p.out_vec8 && p.OC % 8 == 0 && p.pre_scale == nullptr && p.post_scale == nullptr && rows <= p.stat_cap;
    if (ok) *p.stat_written = (int32_t)rows;
    else p.stat_rows = nullptr;
  }
  if (p.a == 2) {                        // strided mode; a mask, if any, stays with the caller's sweep
    if (patch_to(p) == 128) return out_f32 ? launch_one<float, 128, false, true>(p, l.flops, stream) : launch_one<uint16_t, 128, false, true>(p, l.flops, stream);
    return out_f32 ? launch_one<float, 64, false, true>(p, l.flops, stream) : launch_one<uint16_t, 64, false, true>(p, l.flops, stream);
  }
  const bool mask = p.mask != nullptr && p.out_vec8 && p.OC % 8 == 0 && (reinterpret_cast<uintptr_t>(p.mask) & 15u) == 0;
  int rc;
  if (patch_to(p) == 128) rc = out_f32 ? launch_mask<float, 128>(p, l.flops, mask, stream) : launch_mask<uint16_t, 128>(p, l.flops, mask, stream);
  else rc = out_f32 ? launch_mask<float, 64>(p, l.flops, mask, stream) : launch_mask<uint16_t, 64>(p, l.flops, mask, stream);
  if (mask) p.mask = nullptr;
  return rc;
}

}  // namespace gc
