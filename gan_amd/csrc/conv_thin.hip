// Weights-resident gather convolution for stride-1 gathers over at most 64 input channels (one class, up to nine taps in a
// row-major grid, 64 output channels per workgroup): the 3x3 layers on the largest images of a progressive GAN, where the
// reduction is so short (18 steps of 32 channels) that the patch-resident kernel (conv_patch.hip) spends as long on a
// workgroup's fixed parts -- first patch, weight stream start-up, 64 KiB of stores -- as on its MFMAs.
//
// One persistent workgroup of four waves per CU.  It brings the WHOLE filter of its 64 output channels into LDS once
// ([tap][half of the 64 input channels][64 oc][32 ch], 2 * T slots of 4 KiB, at most 72 KiB) and then walks 8 x 32
// rectangles of the output: per rectangle two patch chunks of 32 channels (22 KiB each, the patch-resident kernel's
// layout and swizzle) through a ring of three buffers.  While chunk k is multiplied, chunk k + 1 has landed or is landing
// and chunk k + 2 is issued, so a rectangle's patch is in flight one whole rectangle ahead and the previous rectangle's
// stores drain beside the MFMAs: nothing streams per step, there is one barrier per chunk (9 taps x 16 MFMAs per wave)
// instead of one per tap, and no start-up per rectangle.  Loads that must not wait for stores are ISSUED BEFORE them
// (LDS-DMA loads and stores retire in order on one counter): the wait for chunk k names the younger chunk and the
// previous rectangle's stores.
//
// Workgroup b runs on XCD b % 8; the workgroups of one rectangle sequence that differ in their output-channel tile are b
// and b + 8, so a patch is fetched from HBM once per XCD.  Wave w computes all 64 channels for a block of 4 rows x 16
// columns of the rectangle and walks the taps column by column: the pixel fragments of patch rows y .. y + 5 at column
// shift s serve all three taps (r, s) -- tap r of output row i reads patch row i + r -- so a filter column costs 6 pixel
// and 12 weight fragment reads for 48 MFMAs (the LDS's read rate, 128 bytes per clock, is what a 64-channel tile runs
// against: 8 reads per 16 MFMAs would keep it busy all the time).  The reads of the next tap are issued in front of the
// MFMAs of the current one (two register sets).
#include <stdio.h>

#include "conv_gather.h"

namespace gc {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr uint32_t kOutOfRange = 0x80000000u;
constexpr int kTH = 8, kTW = 32;
constexpr int kNW = 4;
constexpr int kPInstr = 22;                      // patch DMA instructions of 16 pixels per chunk
constexpr int kMaxPatch = kPInstr * 16;
constexpr int kPatchBytes = kPInstr * 1024;
constexpr int kPatchBufs = 3;
constexpr int kMaxT = 9;
constexpr int kSlotBytes = 64 * 32 * 2;          // [64 oc][32 ch]
constexpr int kWBytes = 2 * kMaxT * kSlotBytes;
constexpr int kSmem = kWBytes + kPatchBufs * kPatchBytes;

struct ThinGeom {                                // launch-invariant, worked out on the host
  int32_t T, nw;                                 // taps, taps per filter row
  int32_t dh_min, dw_min, PH, PW;                // patch origin relative to the rectangle, patch extent
  int32_t to0, dto_r, dto_s;                     // patch pixel offset of tap (r, s) = to0 + r * dto_r + s * dto_s
  int32_t wt0, dwt_r, dwt_s;                     // its filter tap index, likewise
  int32_t n_rects, n_seq, tiles_w, tiles_h;
  int32_t rflip, sflip;                          // the tap grid runs against the patch rows / columns (a data gradient's flipped filter)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

// channel (relative to the tile) held by the weight row at LDS row `row` (as in conv_patch.hip): a lane's two MFMA blocks
// of a group of 32 hold 8 consecutive channels of a pixel
__device__ __forceinline__ int oc_of_row(int row) {
  const int f = (row >> 4) & 1, g = (row >> 2) & 3, e = row & 3;
  return (row & ~31) + 8 * g + 4 * f + e;
}

template <typename TOut, bool MASK>
__global__ __launch_bounds__(kNW * 64, 1) void conv_thin_kernel(const GatherParams p, const ThinGeom q) {
  constexpr int FO = 4;
  constexpr int NS = sizeof(TOut) == 4 ? 16 : 8;           // store instructions per lane and rectangle
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* w_base = smem;
  char* patch_base = smem + kWBytes;

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, c16 = lane & 15;

  const int xcd = blockIdx.x & 7, grp = blockIdx.x >> 3;
  const int oc0 = (grp % p.n_oc_tiles) * 64;
  const int seq = xcd + 8 * (grp / p.n_oc_tiles);
  const int nr = seq < q.n_rects ? (q.n_rects - seq + q.n_seq - 1) / q.n_seq : 0;
  if (nr == 0) return;
  const int T = q.T, PW = q.PW, npatch = q.PH * q.PW;

  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, (int)p.w_bytes, 0x00020000);

  // ---- the filter: slot 2 * tap + half, one DMA instruction per wave and slot
  {
    const int row = wave * 16 + (lane >> 2), pos = lane & 3;
    const int oc = oc0 + oc_of_row(row);
    const uint32_t voff = oc < p.OC ? (uint32_t)(((size_t)oc * p.row_pitch + (pos ^ ((row >> 1) & 3)) * 8) * 2) : kOutOfRange;
    int tr = 0, ts = 0;
    for (int tap = 0; tap < T; tap++) {
      const int wt = q.wt0 + tr * q.dwt_r + ts * q.dwt_s;
#pragma unroll
      for (int kh = 0; kh < 2; kh++)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(w_base + (tap * 2 + kh) * kSlotBytes + wave * 1024), 16, (int)voff,
                                                 (wt * p.ICp + kh * 32) * 2, 0, 0);
      if (++ts == q.nw) { ts = 0; tr++; }
    }
  }

  // ---- patch DMA: instruction j * 4 + wave covers patch pixels 16 (j * 4 + wave) .. + 15 (waves 0, 1: six; waves 2, 3: five)
  const int npi = wave < 2 ? 6 : 5;
  int p_r[6], p_c[6], p_chunk[6];
  bool p_in[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int px = 16 * (j * kNW + wave) + (lane >> 2);
    p_chunk[j] = (lane & 3) ^ ((px >> 1) & 3);
    p_r[j] = px / PW;
    p_c[j] = px - p_r[j] * PW;
    p_in[j] = px < npatch;
  }
  // rectangle (sample, row tile, column tile) of a sequence position: cursors that advance by n_seq rectangles with carries
  struct Rect { int n, th, tw; };
  auto rect_at = [&](int j) {
    const int i = seq + j * q.n_seq;
    Rect r;
    r.tw = i % q.tiles_w;
    const int rest = i / q.tiles_w;
    r.th = rest % q.tiles_h;
    r.n = rest / q.tiles_h;
    return r;
  };
  const int step_tw = q.n_seq % q.tiles_w, step_rest = q.n_seq / q.tiles_w;
  const int step_th = step_rest % q.tiles_h, step_n = step_rest / q.tiles_h;
  auto advance = [&](Rect& r) {
    r.tw += step_tw;
    int c = r.tw >= q.tiles_w ? 1 : 0;
    r.tw -= c * q.tiles_w;
    r.th += step_th + c;
    c = r.th >= q.tiles_h ? 1 : 0;
    r.th -= c * q.tiles_h;
    r.n += step_n + c;
  };
  // per-lane part of a patch load's address: (patch row, patch column, 16-byte chunk) relative to the patch origin
  int p_lane[6];
#pragma unroll
  for (int j = 0; j < 6; j++) p_lane[j] = ((p_r[j] * p.IW + p_c[j]) * p.in_cstride + p.in_coff + p_chunk[j] * 8) * 2;
  // chunk (rectangle j, channel half kh) into ring buffer `buf`; live = false: every lane out of range (zeros into a
  // buffer nobody reads, so that the counted waits stay uniform)
  auto issue_patch = [&](const Rect& rc, int kh, int buf, bool live) {
    char* pb = patch_base + buf * kPatchBytes;
    const int ih0 = rc.th * kTH + q.dh_min, iw0 = rc.tw * kTW + q.dw_min;
    const uint32_t base = (uint32_t)((((size_t)rc.n * p.IH + ih0) * p.IW + iw0) * p.in_cstride) * 2u;     // (may wrap below zero at an edge: masked there)
    // a rectangle whose patch lies inside the image needs no per-lane bounds checks
    const bool inside = ih0 >= 0 && iw0 >= 0 && ih0 + q.PH <= p.IH && iw0 + q.PW <= p.IW;
#pragma unroll
    for (int jj = 0; jj < 6; jj++)
      if (jj < 5 || npi == 6) {
        bool ok = live & p_in[jj] & (kh * 32 + p_chunk[jj] * 8 < p.IC);
        if (!inside) {
          const int ih = ih0 + p_r[jj], iw = iw0 + p_c[jj];
          ok = ok & ((unsigned)ih < (unsigned)p.IH) & ((unsigned)iw < (unsigned)p.IW);
        }
        // (the scalar offset stays uniform: an out-of-range lane offset stays out of range with it added)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(pb + (jj * kNW + wave) * 1024), 16,
                                                 (int)(ok ? base + (uint32_t)p_lane[jj] : kOutOfRange), kh * 64, 0, 0);
      }
  };

  // ---- fragment addressing (layouts and swizzles of conv_patch.hip): patch rows y0 .. y0 + 5 at this wave's 16 columns, weight blocks
  const int row0 = 4 * (wave >> 1), col0 = 16 * (wave & 1);
  int base_idx[6];
#pragma unroll
  for (int i = 0; i < 6; i++) base_idx[i] = (row0 + i) * PW + col0 + c16;
  int a_addr[FO];
#pragma unroll
  for (int ob = 0; ob < FO; ob++) {
    const int row = ob * 16 + c16;
    a_addr[ob] = row * 64 + ((g ^ ((row >> 1) & 3)) << 4);
  }
  auto load_b = [&](const char* pb, int sshift, bf16x8 (&b)[6]) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int idx = base_idx[i] + sshift;
      b[i] = *reinterpret_cast<const bf16x8*>(pb + idx * 64 + ((g ^ ((idx >> 1) & 3)) << 4));
    }
  };
  auto load_a = [&](int kh, int tap, bf16x8 (&a)[FO]) {
    const char* wb = w_base + (tap * 2 + kh) * kSlotBytes;
#pragma unroll
    for (int ob = 0; ob < FO; ob++) a[ob] = *reinterpret_cast<const bf16x8*>(wb + a_addr[ob]);
  };

  f32x4 acc[FO][4];
  // accumulators start at the bias (accumulator row r of block ob is channel 32 (ob >> 1) + 8 g + 4 (ob & 1) + r), or at zero
  // when a per-sample scale stands between the sums and the bias
  auto clear_acc = [&](const float (&bv)[2][8], bool bias_first) {
#pragma unroll
    for (int i = 0; i < FO; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc[i][j][r] = bias_first ? bv[i >> 1][4 * (i & 1) + r] : 0.f;
  };
  // all nine taps of one chunk out of patch buffer pb, straight-line, patch column shift by column shift; the tap that reads
  // patch offset (dr, dc) sits at grid position (dr or 2 - dr, dc or 2 - dc) of the filter slots
  auto slot_of = [&](int dr, int dc) { return (q.rflip ? 2 - dr : dr) * 3 + (q.sflip ? 2 - dc : dc); };
  auto compute = [&](const char* pb, int kh) {
    bf16x8 a[2][FO], b[2][6];
    load_b(pb, 0, b[0]);
    load_a(kh, slot_of(0, 0), a[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int dc = 0; dc < 3; dc++) {
#pragma unroll
      for (int dr = 0; dr < 3; dr++) {
        const int step = dc * 3 + dr;
        if (step + 1 < 9) load_a(kh, slot_of((step + 1) % 3, (step + 1) / 3), a[(step + 1) & 1]);
        if (dr == 0 && dc + 1 < 3) load_b(pb, dc + 1, b[(dc + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ob = 0; ob < FO; ob++)
#pragma unroll
          for (int pg = 0; pg < 4; pg++)
            acc[ob][pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[step & 1][ob], b[dc & 1][pg + dr], acc[ob][pg], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // bias of this lane's channels (two groups of 8), loaded once
  float bias_v[2][8];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias != nullptr) {
      b0 = *reinterpret_cast<const f32x4*>(p.bias + oc0 + 32 * m + 8 * g);
      b1 = *reinterpret_cast<const f32x4*>(p.bias + oc0 + 32 * m + 8 * g + 4);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) { bias_v[m][r] = b0[r]; bias_v[m][4 + r] = b1[r]; }
  }
  const float ls = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
  TOut* out = reinterpret_cast<TOut*>(p.out);

  const bool bias_first = p.pre_scale == nullptr;
  auto store_rect = [&](const Rect& rc) {
    const int n = rc.n, th = rc.th, tw = rc.tw;
#pragma unroll
    for (int pg = 0; pg < 4; pg++) {
      const int oh = th * kTH + row0 + pg, ow = tw * kTW + col0 + c16;
      const size_t pix = ((size_t)n * p.OH + oh) * p.OW + ow;
      TOut* dst = out + pix * p.out_cstride + p.out_coff;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int oc = oc0 + 32 * m + 8 * g;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; r++) { v[r] = acc[2 * m][pg][r]; v[4 + r] = acc[2 * m + 1][pg][r]; }
        if (p.pre_scale != nullptr) {
          const float* sc = p.pre_scale + (size_t)n * p.OC + oc;
          const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
#pragma unroll
          for (int r = 0; r < 4; r++) { v[r] *= s0[r]; v[4 + r] *= s1[r]; }
        }
        if (!bias_first) {
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] += bias_v[m][r];
        }
        if (ls != 1.f) {
#pragma unroll
          for (int r = 0; r < 8; r++) v[r] = fmaxf(v[r], v[r] * ls);
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
            const u32x4 mq = *reinterpret_cast<const u32x4*>(mk);
#pragma unroll
            for (int r = 0; r < 4; r++) {
              mv[2 * r] = __builtin_bit_cast(float, mq[r] << 16);
              mv[2 * r + 1] = __builtin_bit_cast(float, mq[r] & 0xffff0000u);
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
      }
    }
  };

  // ---- chunks k = 2 j + kh through ring buffer k % 3.  Younger than chunk k's loads when chunk k is waited for: chunk
  // k + 1's loads, and from the second rectangle on the previous rectangle's stores.
  Rect cur = rect_at(0), nxt = cur;
  issue_patch(cur, 0, 0, true);
  issue_patch(cur, 1, 1, true);
  advance(nxt);
  int buf = 0;                                   // ring buffer of the chunk being multiplied
  for (int j = 0; j < nr; j++) {
    clear_acc(bias_v, bias_first);
#pragma unroll
    for (int kh = 0; kh < 2; kh++) {
      if (j == 0) {
        if (npi == 6) wait_vmcnt<6>(); else wait_vmcnt<5>();
      } else {
        if (npi == 6) wait_vmcnt<6 + NS>(); else wait_vmcnt<5 + NS>();
      }
      __builtin_amdgcn_s_barrier();              // chunk k has landed for every wave; every wave is done with chunk k - 1
      __builtin_amdgcn_sched_barrier(0);
      const int nbuf = buf == 0 ? 2 : buf - 1;   // (k + 2) % 3 = (k - 1) % 3
      issue_patch(nxt, kh, nbuf, j + 1 < nr);
      __builtin_amdgcn_sched_barrier(0);
      compute(patch_base + buf * kPatchBytes, kh);
      buf = buf == 2 ? 0 : buf + 1;
      __builtin_amdgcn_sched_barrier(0);
    }
    store_rect(cur);
    cur = nxt;
    advance(nxt);
    __builtin_amdgcn_sched_barrier(0);
  }
  wait_vmcnt<0>();                               // the zero-filling tail loads target this workgroup's LDS: let them land
}

template <typename TOut>
int launch_one(const GatherParams& p, const ThinGeom& q, bool mask, unsigned grid, double flops, hipStream_t stream) {
  auto kern = mask ? conv_thin_kernel<TOut, true> : conv_thin_kernel<TOut, false>;
  if (ensure_dynamic_smem(reinterpret_cast<const void*>(kern), kSmem) != GC_OK) return GC_HIP_ERROR;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_thin_kernel<%s, %s>", sizeof(TOut) == 4 ? "float" : "unsigned short", mask ? "true" : "false");
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kNW * 64), kSmem, stream, p, q);
  return check_hip(hipGetLastError(), "conv_thin launch");
}

bool thin_geometry(const GatherParams& p, const GatherLaunch& l, ThinGeom& q, unsigned& grid) {
  static const bool off = getenv("GC_NO_THIN") != nullptr;
  if (off || !l.in_bf16 || l.mode != kVecLoads || p.a != 1 || p.b != 1 || p.tt.n_classes != 1 || p.ICp != 64) return false;
  if (p.OC % 64 != 0 || p.OC > 256 || !p.out_vec8) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.stat_rows != nullptr || p.out_bf16 != nullptr) return false;
  if (!(p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || (p.act == GC_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f))) return false;
  if (p.mask != nullptr && (reinterpret_cast<uintptr_t>(p.mask) & 15u) != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.bias) & 15u) != 0) return false;
  const int T = p.tt.cls_tap_start[1] - p.tt.cls_tap_start[0];
  if (T != kMaxT) return false;                   // 3 x 3 (the multiply loop is unrolled over its nine taps)
  int dh_min = 127, dh_max = -128, dw_min = 127, dw_max = -128;
  for (int e = 0; e < T; e++) {
    dh_min = p.tt.dh[e] < dh_min ? p.tt.dh[e] : dh_min; dh_max = p.tt.dh[e] > dh_max ? p.tt.dh[e] : dh_max;
    dw_min = p.tt.dw[e] < dw_min ? p.tt.dw[e] : dw_min; dw_max = p.tt.dw[e] > dw_max ? p.tt.dw[e] : dw_max;
  }
  const int nw = dw_max - dw_min + 1, nh = dh_max - dh_min + 1;
  if (nw != 3 || nh != 3) return false;
  q.T = T; q.nw = nw; q.dh_min = dh_min; q.dw_min = dw_min;
  q.PH = kTH + dh_max - dh_min; q.PW = kTW + dw_max - dw_min;
  if (q.PH * q.PW > kMaxPatch) return false;
  auto to_of = [&](int e) { return (p.tt.dh[e] - dh_min) * q.PW + (p.tt.dw[e] - dw_min); };
  q.to0 = to_of(0); q.wt0 = p.tt.wtap[0];
  q.dto_s = nw > 1 ? to_of(1) - q.to0 : 0; q.dwt_s = nw > 1 ? (int)p.tt.wtap[1] - q.wt0 : 0;
  q.dto_r = nh > 1 ? to_of(nw) - q.to0 : 0; q.dwt_r = nh > 1 ? (int)p.tt.wtap[nw] - q.wt0 : 0;
  // unit steps in either direction: the tap at grid row r reads patch row i + r (or i + 2 - r) of output row i
  if ((q.dto_r != q.PW && q.dto_r != -q.PW) || (q.dto_s != 1 && q.dto_s != -1)) return false;
  q.rflip = q.dto_r < 0; q.sflip = q.dto_s < 0;
  for (int i = 0; i < T; i++) {          // row-major grid, offsets and filter taps affine in (row, column)
    const int r = i / nw, s = i % nw;
    if (to_of(i) != q.to0 + r * q.dto_r + s * q.dto_s || (int)p.tt.wtap[i] != q.wt0 + r * q.dwt_r + s * q.dwt_s) return false;
  }
  const int oh = p.tt.cls_oh[0], ow = p.tt.cls_ow[0];
  if (oh % kTH != 0 || ow % kTW != 0 || oh != p.OH || ow != p.OW) return false;
  q.tiles_w = ow / kTW; q.tiles_h = oh / kTH;
  const long rects = (long)p.N * q.tiles_w * q.tiles_h;
  const int n_oc = p.OC / 64;
  const int cus = device_cu_count();
  const int per_xcd = cus / 8 / n_oc;            // rectangle sequences per XCD
  if (per_xcd < 1 || cus % 8 != 0) return false;
  q.n_seq = 8 * per_xcd;
  if (rects < 8l * q.n_seq || rects >= (1l << 30)) return false;     // at least eight rectangles per workgroup to pay for the filter load
  q.n_rects = (int)rects;
  grid = (unsigned)(q.n_seq * n_oc);
  return true;
}

}  // namespace

bool plan_thin(const GatherParams& p, const GatherLaunch& l) {
  ThinGeom q;
  unsigned grid;
  return thin_geometry(p, l, q, grid);
}

// Applies p.mask itself (and clears it).
int launch_thin(GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  ThinGeom q;
  unsigned grid;
  if (!thin_geometry(p, l, q, grid)) return GC_UNSUPPORTED_GEOMETRY;
  p.n_oc_tiles = p.OC / 64;
  const bool mask = p.mask != nullptr;
  const int rc = out_f32 ? launch_one<float>(p, q, mask, grid, l.flops, stream) : launch_one<uint16_t>(p, q, mask, grid, l.flops, stream);
  p.mask = nullptr;
  return rc;
}

}  // namespace gc
