// Patch-resident gather convolution for stride-1 gathers with several taps per class: 3x3 stride-1 forward / data
// gradient, and the 2x2-tap parity classes of a 4x4 stride-2 data gradient (= transposed-conv forward).
//
// The tiled LDS-DMA kernel (conv_gather_dma.hip) re-loads a tile's pixels once per tap: a 3x3 layer moves every input
// element into LDS nine times, and on large layers that traffic (L2 -> LDS at about 40 GB/s per CU chip-wide), not the
// matrix cores, sets the pace.  Here a workgroup owns a 16 x 32 rectangle of one sample's (class-local) output and
// 128 output channels.  Per 32-channel chunk it loads the rectangle's input PATCH once ((16 + dh range) x (32 + dw
// range) pixels, zeros outside the image) and forms the B operand of every tap from it at a shifted LDS address; only
// the weights ([128 oc][32 ch] per tap, 8 KiB) stream per step.  For 3x3 that is 39 + 9 * 8 = 111 KiB per chunk
// instead of 9 * (32 + 8) = 360 KiB on a 128 x 512 tile: 3.2x fewer bytes into LDS per FLOP.
//
// Pipeline: a step is one chunk x TS taps (TS = 3 for nine taps, 2 for four: 96 / 64 MFMAs per wave between two
// barriers); weights go through a 3-slot ring (two steps in flight), the patch is double-buffered and the next
// chunk's patch is issued at the first step of the current one.  All loads are LDS-DMA
// with buffer addressing (per-lane offsets fixed, chunk / tap moves only the scalar offset, masked lanes take an
// out-of-range offset and receive zeros).  LDS rows are 64 bytes (32 channels); the 16-byte chunk position is XORed
// with (row >> 1) & 3 on the source side: with the lane groups a ds_read_b128 is served in, the fragments of 16
// consecutive rows starting anywhere are then conflict-free (modelled in tools/lds_conflicts.py).
// 8 waves; wave w computes all 128 channels for rows 2w, 2w + 1 of the rectangle (64 pixels): 32 MFMAs per tap
// against 8 + 4 fragment reads.
#include <stdio.h>

#include <mutex>

#include "conv_gather.h"

namespace gc {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr uint32_t kOutOfRange = 0x80000000u;
constexpr int kTH = 16, kTW = 32, kTO = 128, kCK = 32;     // rectangle, channel tile, channels per chunk
constexpr int kNW = 8;                                      // waves
constexpr int kMaxPatch = 640;                              // pixels (40 DMA instructions of 16 pixels)
constexpr int kPJ = 5;                                      // patch DMA instructions per wave per chunk (8 * 5 = 40)
constexpr int kPatchBytes = kNW * kPJ * 1024;               // 40 KiB per buffer
constexpr int kWBytes = kTO * kCK * 2;                      // 8 KiB per tap
constexpr int kWSlots = 3;
constexpr int smem_bytes(int ts) { return 2 * kPatchBytes + kWSlots * ts * kWBytes + kMaxTaps * 4; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

template <typename TOut, int TS>
__global__ __launch_bounds__(kNW * 64) void conv_patch_kernel(const GatherParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kSlot = TS * kWBytes;
  char* patch_base = smem;
  char* w_base = smem + 2 * kPatchBytes;
  int* tap_lds = reinterpret_cast<int*>(w_base + kWSlots * kSlot);

  const int t = threadIdx.x;
  const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = lane >> 4, c16 = lane & 15;

  if (t < kMaxTaps) tap_lds[t] = ((int)p.tt.dh[t] & 0xff) | (((int)p.tt.dw[t] & 0xff) << 8) | ((int)p.tt.wtap[t] << 16);
  __syncthreads();

  // ---- which (class, sample, rectangle, channel tile)
  const uint32_t nblocks = (uint32_t)p.n_pix_tiles * (uint32_t)p.n_oc_tiles;
  const uint32_t lid = xcd_remap(blockIdx.x, nblocks);
  const int pix_tile = lid / p.n_oc_tiles;
  const int oc0 = (lid % p.n_oc_tiles) * kTO;
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
  const int T = tap_end - tap_begin;
  int dh_min = 127, dh_max = -128, dw_min = 127, dw_max = -128;
  for (int e = tap_begin; e < tap_end; e++) {
    const int v = __builtin_amdgcn_readfirstlane(tap_lds[e]);
    const int dh = (int)(int8_t)(v & 0xff), dw = (int)(int8_t)((v >> 8) & 0xff);
    dh_min = min(dh_min, dh); dh_max = max(dh_max, dh);
    dw_min = min(dw_min, dw); dw_max = max(dw_max, dw);
  }
  const int PH = kTH + dh_max - dh_min, PW = kTW + dw_max - dw_min;
  const int npatch = PH * PW;

  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, (int)p.w_bytes, 0x00020000);

  // ---- per-lane source offsets: patch instruction j of this wave covers patch pixels 16 (j * 8 + wave) .. + 15
  uint32_t a_off[kPJ];
  int a_chunk[kPJ];
#pragma unroll
  for (int j = 0; j < kPJ; j++) {
    const int px = 16 * (j * kNW + wave) + (lane >> 2);
    const int pos = lane & 3;
    const int chunk = pos ^ ((px >> 1) & 3);
    const int pr = px / PW, pc = px - pr * PW;
    const int ih = th_ * kTH + pr + dh_min, iw = tw_ * kTW + pc + dw_min;
    const bool ok = (px < npatch) & ((unsigned)ih < (unsigned)p.IH) & ((unsigned)iw < (unsigned)p.IW);
    a_chunk[j] = ok ? chunk * 8 : (1 << 30);     // a masked lane fails the channel check of every chunk
    a_off[j] = (uint32_t)((((size_t)n * p.IH + (ok ? ih : 0)) * p.IW + (ok ? iw : 0)) * p.in_cstride + p.in_coff + chunk * 8) * 2u;
  }
  uint32_t w_voff;
  {
    const int row = wave * 16 + (lane >> 2), pos = lane & 3;
    const int oc = oc0 + row;
    w_voff = oc < p.OC ? (uint32_t)(((size_t)oc * p.row_pitch + (pos ^ ((row >> 1) & 3)) * 8) * 2) : kOutOfRange;
  }

  const int nchunks = p.ICp / kCK;
  const int spc = T / TS;                 // steps per chunk
  const int nsteps = nchunks * spc;

  auto issue_patch = [&](int c) {
    char* pb = patch_base + (c & 1) * kPatchBytes;
    const int c0 = c * kCK;
#pragma unroll
    for (int j = 0; j < kPJ; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(pb + (j * kNW + wave) * 1024), 16,
                                               (int)((c0 + a_chunk[j]) < p.IC ? a_off[j] : kOutOfRange), c0 * 2, 0, 0);
  };
  // weights of step (chunk c, tap group gi): TS taps of [128 oc][32 ch]
  auto issue_w = [&](int slot, int c, int gi) {
#pragma unroll
    for (int k = 0; k < TS; k++) {
      const int e = __builtin_amdgcn_readfirstlane(tap_lds[tap_begin + gi * TS + k]);
      const uint32_t soff = (uint32_t)(((e >> 16) * p.ICp + c * kCK) * 2);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(w_base + slot * kSlot + k * kWBytes + wave * 1024), 16, (int)w_voff,
                                               (int)soff, 0, 0);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // patch index of this lane's pixel in each of its four 16-pixel groups (tap offset (dh_min, dw_min))
  int base_idx[4];
#pragma unroll
  for (int pg = 0; pg < 4; pg++) base_idx[pg] = (2 * wave + (pg >> 1)) * PW + (pg & 1) * 16 + c16;
  // weight fragment offsets inside a tap's [128][32] tile
  int a_addr[8];
#pragma unroll
  for (int ob = 0; ob < 8; ob++) {
    const int row = ob * 16 + c16;
    a_addr[ob] = row * 64 + ((g ^ ((row >> 1) & 3)) << 4);
  }

  auto compute = [&](int slot, int c, int gi) {
    const char* pb = patch_base + (c & 1) * kPatchBytes;
#pragma unroll
    for (int k = 0; k < TS; k++) {
      const char* wb = w_base + slot * kSlot + k * kWBytes;
      const int e = __builtin_amdgcn_readfirstlane(tap_lds[tap_begin + gi * TS + k]);
      const int toff = ((int)(int8_t)(e & 0xff) - dh_min) * PW + ((int)(int8_t)((e >> 8) & 0xff) - dw_min);
      bf16x8 b[4], a[8];
#pragma unroll
      for (int pg = 0; pg < 4; pg++) {
        const int idx = base_idx[pg] + toff;
        b[pg] = *reinterpret_cast<const bf16x8*>(pb + idx * 64 + ((g ^ ((idx >> 1) & 3)) << 4));
      }
#pragma unroll
      for (int ob = 0; ob < 8; ob++) a[ob] = *reinterpret_cast<const bf16x8*>(wb + a_addr[ob]);
#pragma unroll
      for (int ob = 0; ob < 8; ob++)
#pragma unroll
        for (int pg = 0; pg < 4; pg++) acc[ob][pg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ob], b[pg], acc[ob][pg], 0, 0, 0);
    }
  };

  // ---- prologue: patch 0, then the weights of steps 0 and 1
  issue_patch(0);
  int ic = 0, igi = 0, islot = 0;     // (chunk, tap group, ring slot) of the next weight step to issue
  int issued = 0;
  for (int s = 0; s < kWSlots - 1 && issued < nsteps; s++) {
    issue_w(islot, ic, igi);
    issued++;
    islot = islot + 1 == kWSlots ? 0 : islot + 1;
    if (++igi == spc) { igi = 0; ic++; }
  }
  // ---- steps
  int cc = 0, cgi = 0, cslot = 0;     // (chunk, tap group, ring slot) of the step being computed
  for (int s = 0; s < nsteps; s++) {
    // Loads younger than the weights of step s: the next step's weights, plus the next chunk's patch when it went out
    // at the previous step (step 0 of this chunk).  The patch of THIS chunk went out a whole chunk earlier: older.
    if (s + 1 >= nsteps) wait_vmcnt<0>();
    else if (cgi == 1 && cc + 1 < nchunks) wait_vmcnt<TS + kPJ>();
    else wait_vmcnt<TS>();
    __builtin_amdgcn_s_barrier();        // step s has landed for every wave; the previous step's slot and the other patch buffer are free
    __builtin_amdgcn_sched_barrier(0);
    if (cgi == 0 && cc + 1 < nchunks) issue_patch(cc + 1);
    if (issued < nsteps) {
      issue_w(islot, ic, igi);
      issued++;
      islot = islot + 1 == kWSlots ? 0 : islot + 1;
      if (++igi == spc) { igi = 0; ic++; }
    }
    compute(cslot, cc, cgi);
    cslot = cslot + 1 == kWSlots ? 0 : cslot + 1;
    if (++cgi == spc) { cgi = 0; cc++; }
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: lane owns 4 consecutive channels of one pixel per (channel block, pixel group)
  TOut* out = reinterpret_cast<TOut*>(p.out);
  const bool vec_ok = p.out_vec != 0;
#pragma unroll
  for (int pg = 0; pg < 4; pg++) {
    const int oh = th_ * kTH + 2 * wave + (pg >> 1), ow = tw_ * kTW + (pg & 1) * 16 + c16;
    if (oh >= OHc || ow >= OWc) continue;
    const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
    TOut* dst = out + pix * p.out_cstride + p.out_coff;
#pragma unroll
    for (int ob = 0; ob < 8; ob++) {
      const int oc = oc0 + ob * 16 + g * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float x = acc[ob][pg][r];
        if (p.bias != nullptr && oc + r < p.OC) x += p.bias[oc + r];
        v[r] = apply_act(x, p.act, p.slope);
      }
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
}

template <typename TOut, int TS>
int launch_one(const GatherParams& p, double flops, hipStream_t stream) {
  constexpr int kSmem = smem_bytes(TS);
  auto kern = conv_patch_kernel<TOut, TS>;
  static std::once_flag once;            // (launches come from several host threads: the autograd engine's workers)
  static hipError_t attr_err = hipSuccess;
  std::call_once(once, [&]() {
    attr_err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kSmem);
  });
  if (check_hip(attr_err, "hipFuncSetAttribute")) return GC_HIP_ERROR;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_patch_kernel<%s, %d>", sizeof(TOut) == 4 ? "float" : "unsigned short", TS);
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)p.n_pix_tiles * (unsigned)p.n_oc_tiles), dim3(kNW * 64), kSmem, stream, p);
  return check_hip(hipGetLastError(), "conv_patch launch");
}

}  // namespace

// Decides whether the patch kernel takes the problem and, if so, replaces the tile bookkeeping in p (class tile
// prefix, tile counts) by its own.  The caller has established: bf16 input, 16-byte aligned channel slices, both
// tensors below 2 GiB, no reduction split.
bool plan_patch(GatherParams& p, const GatherLaunch& l) {
  if (!l.in_bf16 || l.mode != kVecLoads || p.a != 1 || p.OC < kTO || p.ICp % kCK != 0) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.out_bf16 != nullptr) return false;
  long tiles = 0;
  int starts[kMaxClasses + 1];
  for (int c = 0; c < p.tt.n_classes; c++) {
    const int t0 = p.tt.cls_tap_start[c], t1 = p.tt.cls_tap_start[c + 1];
    const int ts = (p.tt.cls_tap_start[1] - p.tt.cls_tap_start[0]) % 3 == 0 ? 3 : 2;   // taps per step, from class 0
    if ((t1 - t0) % ts != 0 || (t1 - t0) / ts < 2) return false;   // the wait counts assume at least two steps per chunk
    int dh_min = 127, dh_max = -128, dw_min = 127, dw_max = -128;
    for (int e = t0; e < t1; e++) {
      dh_min = p.tt.dh[e] < dh_min ? p.tt.dh[e] : dh_min; dh_max = p.tt.dh[e] > dh_max ? p.tt.dh[e] : dh_max;
      dw_min = p.tt.dw[e] < dw_min ? p.tt.dw[e] : dw_min; dw_max = p.tt.dw[e] > dw_max ? p.tt.dw[e] : dw_max;
    }
    if ((kTH + dh_max - dh_min) * (kTW + dw_max - dw_min) > kMaxPatch) return false;
    const int oh = p.tt.cls_oh[c], ow = p.tt.cls_ow[c];
    if (oh % kTH != 0 || ow % kTW != 0) return false;    // partial rectangles would idle most of a workgroup
    starts[c] = (int)tiles;
    tiles += (long)p.N * (oh / kTH) * (ow / kTW);
  }
  const int n_oc_tiles = (p.OC + kTO - 1) / kTO;
  if (tiles * n_oc_tiles < 512 || tiles * n_oc_tiles >= (1l << 31)) return false;   // fewer: the tiled kernels fill the chip better
  for (int c = 0; c < p.tt.n_classes; c++) p.tt.cls_tile_start[c] = starts[c];
  for (int c = p.tt.n_classes; c <= kMaxClasses; c++) p.tt.cls_tile_start[c] = (int)tiles;
  p.n_pix_tiles = (int)tiles;
  p.n_oc_tiles = n_oc_tiles;
  p.cls_interleave = 0;
  return true;
}

int launch_patch(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  if (p.n_pix_tiles == 0 || p.n_oc_tiles == 0) return GC_OK;
  if ((p.tt.cls_tap_start[1] - p.tt.cls_tap_start[0]) % 3 == 0)
    return out_f32 ? launch_one<float, 3>(p, l.flops, stream) : launch_one<uint16_t, 3>(p, l.flops, stream);
  return out_f32 ? launch_one<float, 2>(p, l.flops, stream) : launch_one<uint16_t, 2>(p, l.flops, stream);
}

}  // namespace gc
