// First layer of the DCGAN / WGAN-GP critic: Conv2d(<= 4 channels -> up to 64, kernel 4, stride 2, padding 1) on fp32
// images (reference BasicGANs/models.py:52-56,77).  1.6 GFLOP against 90 MB of traffic at batch 512: HBM-bound, so the
// goal is to touch every input byte once and to keep the instruction count per output pixel low.
//
// A workgroup owns a 4 x 32 rectangle of one sample's output.  It stages the 10 x 66 input patch into LDS as bf16
// with the channels padded to 4 (8 bytes per pixel, zeros outside the image), so that the 4 taps x 4 channels of one
// filter row are 32 contiguous bytes for every output pixel: the B operand of a 16x16x32 MFMA (two filter rows) is
// one 16-byte LDS read per lane, with no address arithmetic per tap.  The packed weights ([oc][tap][4 channels],
// 64 reduction elements per row) stay in registers as A operands.
#include <stdlib.h>

#include "conv_gather.h"

namespace gc {
namespace {

constexpr int kTH = 4, kTW = 32;                 // output rectangle of a workgroup (a multiple of 4 rows: one or more per wave)

// S = 2: the 4 x 4 / stride 2 / padding 1 layer above.  S = 1: a 3 x 3 / stride 1 / padding 1 layer on an image (BigGAN's
// first discriminator block, BigGAN/model.py:181-200 on [B, 3, 128, 128]) with the same machinery: the filter rows are
// padded to four taps (the fourth and a fourth row hold zeros), so a reduction step is still two filter rows of 4 taps x 4
// channels, and the patch gets two columns and one row of zeros beyond what the taps with real weights read.
template <int S>
struct ImageGeom {
  static constexpr int K = S == 2 ? 4 : 3;
  static constexpr int VH = S * kTH + K - S, VW = S * kTW + K - S;      // the part of the patch that holds image data
  static constexpr int PH = S == 2 ? VH : VH + 1, PW = S == 2 ? VW : VW + 2;
};

// LEAN: the epilogue's activation is none / ReLU / LeakyReLU with a slope in [0, 1], i.e. max(v, s v) with s = 1 / 0 / slope
//
// A workgroup walks a run of consecutive rectangles (units: sample-major, then rows of rectangles, then columns): the
// weights and the bias are fetched once, and the image loads of rectangle i + 1 are in flight while rectangle i is
// multiplied and stored (two patch buffers, the raw floats wait in registers).
// NB: 16-channel blocks the kernel is built for (4: up to 64 output channels; 8: up to 128, at twice the registers)
// ICP: channels of a patch pixel (4, or 8 for 5 to 8 input channels -- the PatchGAN's image pair: a filter row per MFMA)
// MASK: the stored value is multiplied by the derivative of the ReLU / LeakyReLU whose output is p.mask (gc_epilogue.grad_mask:
// the second-order pass of the critic's first layer, whose result the next layer's masked data gradient would otherwise sweep)
template <typename TOut, bool LEAN, int NB, int ICP, bool MASK = false, int S = 2>
__global__ __launch_bounds__(kThreads) void conv_image_kernel(const GatherParams p, int units, int units_per_wg) {
  static_assert(S == 2 || (S == 1 && ICP == 4 && !MASK), "the stride-1 form: 4-channel patch pixels, no mask");
  constexpr int NKS = ICP / 2;        // reduction steps of 32: 16 taps x ICP channels
  constexpr int kPH = ImageGeom<S>::PH, kPW = ImageGeom<S>::PW;
  __shared__ __attribute__((aligned(16))) uint16_t patch[2][kPH * kPW * ICP];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, c16 = lane & 15;
  const int tiles_w = p.OW / kTW, tiles = tiles_w * (p.OH / kTH);
  const int u0 = (int)blockIdx.x * units_per_wg, u1 = min(units, u0 + units_per_wg);
  if (u0 >= u1) return;

  // weights: A fragments of up to four 16-channel blocks, two reduction steps (filter rows 0-1, 2-3).  The rows of a
  // pair of blocks are interleaved -- row 4 g' + e of block ob is channel 32 (ob / 2) + 8 g' + 4 (ob % 2) + e -- so that a lane's
  // results of the two blocks are 8 consecutive channels of its pixel: one 16-byte store, 64 contiguous bytes per
  // pixel and instruction (a last block without a partner keeps its own order: 4 channels, 8 bytes per lane).
  const int nb = p.OC / 16;
  auto channel_of = [&](int ob, int row) {
    return (ob | 1) < nb ? 32 * (ob >> 1) + 8 * (row >> 2) + 4 * (ob & 1) + (row & 3) : 16 * ob + row;
  };
  bf16x8 wa[NB][NKS];
#pragma unroll
  for (int ob = 0; ob < NB; ob++)
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      u32x4 v = z;
      if (S == 2) {
        if (ob < nb) v = *reinterpret_cast<const u32x4*>(p.wp + (size_t)channel_of(ob, c16) * p.row_pitch + ks * 32 + g * 8);
      } else if (ob < nb) {
        // the packed row holds [3 x 3 taps][4 channels]; lane group g takes filter row 2 ks + g / 2, taps 2 (g % 2) and + 1
        const uint16_t* wrow = p.wp + (size_t)channel_of(ob, c16) * p.row_pitch;
        const int r = 2 * ks + (g >> 1), s0 = 2 * (g & 1);
        if (r < 3) {
          const u32x2 a = *reinterpret_cast<const u32x2*>(wrow + (r * 3 + s0) * 4);
          v[0] = a[0]; v[1] = a[1];
          if (s0 + 1 < 3) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(wrow + (r * 3 + s0 + 1) * 4);
            v[2] = b[0]; v[3] = b[1];
          }
        }
      }
      wa[ob][ks] = __builtin_bit_cast(bf16x8, v);
    }
  f32x4 bias_v[NB];
#pragma unroll
  for (int ob = 0; ob < NB; ob++) {
    bias_v[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr && ob < nb) bias_v[ob] = *reinterpret_cast<const f32x4*>(p.bias + channel_of(ob, 4 * g));
  }
  const float sl = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);

  // ---- the up to three patch pixels of this thread (channels padded to 4)
  constexpr int NP = (kPH * kPW + kThreads - 1) / kThreads;
  int ppr[NP], ppc[NP];
#pragma unroll
  for (int i = 0; i < NP; i++) {
    const int e = t + i * kThreads;
    ppr[i] = e / kPW;
    ppc[i] = e - ppr[i] * kPW;
  }
  f32x4 raw[NP][ICP / 4];
  auto load_unit = [&](int u) {
    const int n = u / tiles, r = u - n * tiles, th = r / tiles_w, tw = r - th * tiles_w;
    const int ih0 = S * kTH * th - 1, iw0 = S * kTW * tw - 1;       // input coordinates of patch element (0, 0)
    const float* img = reinterpret_cast<const float*>(p.in) + (size_t)n * p.IH * p.IW * p.IC;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int ih = ih0 + ppr[i], iw = iw0 + ppc[i];
#pragma unroll
      for (int h = 0; h < ICP / 4; h++) raw[i][h] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t + i * kThreads < kPH * kPW && ppr[i] < ImageGeom<S>::VH && ppc[i] < ImageGeom<S>::VW && (unsigned)ih < (unsigned)p.IH &&
          (unsigned)iw < (unsigned)p.IW) {
        const float* src = img + (uint32_t)((ih * p.IW + iw) * p.IC);
        raw[i][0][0] = src[0];
#pragma unroll
        for (int c = 1; c < ICP; c++)
          if (p.IC > c) raw[i][c >> 2][c & 3] = src[c];
      }
    }
  };
  auto store_unit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NP; i++)
      if (t + i * kThreads < kPH * kPW) {
#pragma unroll
        for (int h = 0; h < ICP / 4; h++)
          *reinterpret_cast<u32x2*>(patch[buf] + (t + i * kThreads) * ICP + 4 * h) = cvt_bf16x4(raw[i][h]);
      }
  };

  load_unit(u0);
  store_unit(0);
  __syncthreads();
  constexpr int RW = kTH / 4;       // wave w: output rows RW * w .. of the rectangle, each in two groups of 16 pixels
  for (int u = u0; u < u1; u++) {
    const int buf = (u - u0) & 1;
    const bool more = u + 1 < u1;
    if (more) load_unit(u + 1);
    const int n = u / tiles, r = u - n * tiles, th = r / tiles_w, tw = r - th * tiles_w;
    const int oh0 = th * kTH, ow0 = tw * kTW;
#pragma unroll
    for (int grp = 0; grp < 2 * RW; grp++) {
      const int oh_l = RW * wave + (grp >> 1), ow_l = (grp & 1) * 16 + c16;
      f32x4 acc[NB];
#pragma unroll
      for (int ob = 0; ob < NB; ob++) acc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) {
        // the 8 reduction elements of lane group g: two pixels of 4 channels (two filter rows per step), or one pixel of 8
        const int pr = ICP == 4 ? S * oh_l + 2 * ks + (g >> 1) : 2 * oh_l + ks, pc = ICP == 4 ? S * ow_l + 2 * (g & 1) : 2 * ow_l + g;
        bf16x8 b;
        if (S == 2) {
          b = *reinterpret_cast<const bf16x8*>(patch[buf] + (pr * kPW + pc) * ICP);
        } else {          // (two neighbouring pixels, 8-byte aligned: the pixel column is not even)
          const u32x2 lo = *reinterpret_cast<const u32x2*>(patch[buf] + (pr * kPW + pc) * ICP);
          const u32x2 hi = *reinterpret_cast<const u32x2*>(patch[buf] + (pr * kPW + pc + 1) * ICP);
          b = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        }
#pragma unroll
        for (int ob = 0; ob < NB; ob++)
          if (ob < nb) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[ob][ks], b, acc[ob], 0, 0, 0);
      }
      const uint32_t pix = (uint32_t)((oh0 + oh_l) * p.OW + ow0 + ow_l);
      const size_t o0 = ((size_t)n * p.OH * p.OW + pix) * p.out_cstride + p.out_coff;
      f32x4 v[NB];
#pragma unroll
      for (int ob = 0; ob < NB; ob++) {
        v[ob] = acc[ob] + bias_v[ob];
        if (LEAN) {
#pragma unroll
          for (int e = 0; e < 4; e++) v[ob][e] = fmaxf(v[ob][e], v[ob][e] * sl);
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) v[ob][e] = apply_act(v[ob][e], p.act, p.slope);
        }
      }
#pragma unroll
      for (int ob = 0; ob < NB; ob += 2) {
        if (ob >= nb) continue;
        if (ob + 1 < nb) {          // a pair of blocks: 8 consecutive channels
          const size_t o = o0 + 32 * (ob >> 1) + 8 * g;
          if (MASK) {
            f32x4 m0, m1;
            if (sizeof(TOut) == 4) {
              m0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.mask) + o);
              m1 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.mask) + o + 4);
            } else {
              const u32x4 q = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.mask) + o);
              m0 = widen_bf16x4(u32x2{q[0], q[1]});
              m1 = widen_bf16x4(u32x2{q[2], q[3]});
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
              v[ob][e] = m0[e] > 0.f ? v[ob][e] : v[ob][e] * p.mask_slope;
              v[ob + 1][e] = m1[e] > 0.f ? v[ob + 1][e] : v[ob + 1][e] * p.mask_slope;
            }
          }
          if (sizeof(TOut) == 4) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v[ob];
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o + 4) = v[ob + 1];
          } else {
            const u32x2 lo = cvt_bf16x4(v[ob]), hi = cvt_bf16x4(v[ob + 1]);
            *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + o) = u32x4{lo[0], lo[1], hi[0], hi[1]};
          }
        } else {
          const size_t o = o0 + 16 * ob + 4 * g;
          if (MASK) {
            f32x4 m0;
            if (sizeof(TOut) == 4) m0 = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.mask) + o);
            else m0 = widen_bf16x4(*reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(p.mask) + o));
#pragma unroll
            for (int e = 0; e < 4; e++) v[ob][e] = m0[e] > 0.f ? v[ob][e] : v[ob][e] * p.mask_slope;
          }
          if (sizeof(TOut) == 4) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o) = v[ob];
          else *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + o) = cvt_bf16x4(v[ob]);
        }
      }
    }
    if (more) store_unit(buf ^ 1);
    __syncthreads();
  }
}

}  // namespace

static bool image_stride1(const GatherParams& p) { return p.T == 9 && p.a == 1; }

// Applies to: fp32 input of at most 4 dense channels (up to 8 with at most 64 output channels), the 4x4 / stride 2 / padding 1 forward pass
// (or, with at most 4 input and 64 output channels, 3x3 / stride 1 / padding 1), up to 128 output
// channels in multiples of 16 (more than 64: the data gradient of a generator's image-producing transposed convolution), output extents that tile by 4 x 32, plain epilogue (bias, activation).
bool plan_image(const GatherParams& p, const GatherLaunch& l) {
  if (l.in_bf16 || l.x3 || l.mode != kThinInput || (p.ICp != 4 && p.ICp != 8) || p.IC > p.ICp || p.in_cstride != p.IC || p.in_coff != 0) return false;
  if (p.ICp == 8 && p.OC > 64) return false;
  if (p.tt.n_classes != 1) return false;
  if (image_stride1(p)) {
    static const bool off = getenv("GC_NO_IMAGE_3X3") != nullptr;
    if (off || p.ICp != 4 || p.OC > 64 || p.row_pitch != 64 || p.IH != p.OH || p.IW != p.OW || p.mask != nullptr) return false;
    for (int e = 0; e < 9; e++)
      if (p.tt.dh[e] != e / 3 - 1 || p.tt.dw[e] != e % 3 - 1 || p.tt.wtap[e] != e) return false;
  } else {
    if (p.T != 16 || p.a != 2 || p.row_pitch != 16 * p.ICp || p.IH != 2 * p.OH || p.IW != 2 * p.OW) return false;
    for (int e = 0; e < 16; e++)
      if (p.tt.dh[e] != e / 4 - 1 || p.tt.dw[e] != e % 4 - 1 || p.tt.wtap[e] != e) return false;
  }
  if (p.OC % 16 != 0 || p.OC > 128 || p.OH % kTH != 0 || p.OW % kTW != 0 || !p.out_vec) return false;
  // 16-byte stores of 8 channels
  if ((reinterpret_cast<uintptr_t>(p.out) & 15u) != 0 || p.out_cstride % 8 != 0 || p.out_coff % 8 != 0) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.out_bf16 != nullptr) return false;
  // (rows of channel statistics on offer are left unwritten -- the norm layer then sweeps the output itself -- except with 5 to
  // 8 input channels, where the general kernel, which writes them, keeps the layer)
  if (p.ICp == 8 && p.stat_rows != nullptr) return false;
  if ((reinterpret_cast<uintptr_t>(p.bias) & 15u) != 0) return false;
  if ((long)p.IH * p.IW * p.IC >= (1l << 31)) return false;      // 32-bit offsets inside a sample
  return (long)p.N * (p.OH / kTH) * (p.OW / kTW) < (1l << 31);
}

// Whether launch_image applies p.mask itself (the caller then clears it): 4-channel patch pixels, up to 64 output channels,
// a 16-byte aligned mask (it shares the output's layout).
bool image_applies_mask(const GatherParams& p) {
  return p.mask != nullptr && !image_stride1(p) && p.ICp == 4 && p.OC <= 64 && p.act == GC_ACT_NONE && (reinterpret_cast<uintptr_t>(p.mask) & 15u) == 0;
}

int launch_image(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  const long units_l = (long)p.N * (p.OH / kTH) * (p.OW / kTW);
  if (units_l == 0) return GC_OK;
  const int units = (int)units_l;
  // about four workgroups per CU, each with at least two rectangles so that the loads of one overlap the other
  static const int wgs = getenv("GC_IMAGE_WGS") ? atoi(getenv("GC_IMAGE_WGS")) : 1024;
  int per = (units + wgs - 1) / wgs;
  if (per < 2) per = units >= 2 ? 2 : 1;
  const unsigned blocks = (unsigned)((units + per - 1) / per);
  ProfScope ps("gc::conv_image_kernel", l.flops, stream);
  const bool lean = p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || (p.act == GC_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f);
  const dim3 grid(blocks);
#define GC_IMAGE_LAUNCH8(T_, L_) hipLaunchKernelGGL((conv_image_kernel<T_, L_, 4, 8>), grid, dim3(kThreads), 0, stream, p, units, per)
#define GC_IMAGE_LAUNCH(T_, L_, NB_) hipLaunchKernelGGL((conv_image_kernel<T_, L_, NB_, 4>), grid, dim3(kThreads), 0, stream, p, units, per)
  if (image_stride1(p)) {
    if (out_f32) {
      if (lean) hipLaunchKernelGGL((conv_image_kernel<float, true, 4, 4, false, 1>), grid, dim3(kThreads), 0, stream, p, units, per);
      else hipLaunchKernelGGL((conv_image_kernel<float, false, 4, 4, false, 1>), grid, dim3(kThreads), 0, stream, p, units, per);
    } else {
      if (lean) hipLaunchKernelGGL((conv_image_kernel<uint16_t, true, 4, 4, false, 1>), grid, dim3(kThreads), 0, stream, p, units, per);
      else hipLaunchKernelGGL((conv_image_kernel<uint16_t, false, 4, 4, false, 1>), grid, dim3(kThreads), 0, stream, p, units, per);
    }
  } else if (image_applies_mask(p)) {
    if (out_f32) hipLaunchKernelGGL((conv_image_kernel<float, true, 4, 4, true>), grid, dim3(kThreads), 0, stream, p, units, per);
    else hipLaunchKernelGGL((conv_image_kernel<uint16_t, true, 4, 4, true>), grid, dim3(kThreads), 0, stream, p, units, per);
  } else if (p.ICp == 8) {
    if (out_f32) { if (lean) GC_IMAGE_LAUNCH8(float, true); else GC_IMAGE_LAUNCH8(float, false); }
    else { if (lean) GC_IMAGE_LAUNCH8(uint16_t, true); else GC_IMAGE_LAUNCH8(uint16_t, false); }
  } else if (p.OC <= 64) {
    if (out_f32) { if (lean) GC_IMAGE_LAUNCH(float, true, 4); else GC_IMAGE_LAUNCH(float, false, 4); }
    else { if (lean) GC_IMAGE_LAUNCH(uint16_t, true, 4); else GC_IMAGE_LAUNCH(uint16_t, false, 4); }
  } else {
    if (out_f32) { if (lean) GC_IMAGE_LAUNCH(float, true, 8); else GC_IMAGE_LAUNCH(float, false, 8); }
    else { if (lean) GC_IMAGE_LAUNCH(uint16_t, true, 8); else GC_IMAGE_LAUNCH(uint16_t, false, 8); }
  }
#undef GC_IMAGE_LAUNCH
#undef GC_IMAGE_LAUNCH8
  return check_hip(hipGetLastError(), "conv_image launch");
}

}  // namespace gc
