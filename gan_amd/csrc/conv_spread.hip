// A stride-1 gather whose input has ONE channel and whose output has many: the data gradient of a score layer that is a
// sliding window, the PatchGAN's Conv2d(512 -> 1, kernel 4, stride 1, padding 1) on 31 x 31 maps (reference
// Pix2Pix/models.py:238; its gradient map is 30 x 30 x 1, the result 31 x 31 x 512).
//
//   out[n][oh][ow][oc] = sum over the taps e of in[n][oh + dh[e]][ow + dw[e]] * w[oc][wtap[e]]   (+ bias, activation)
//
// 2 * T FLOP per output element: bound by writing the result once.  The tiled kernels run it as a GEMM whose reduction (the
// taps of one channel) is padded to 64.  Here a lane owns 8 consecutive output channels and keeps their weights of every
// tap in registers (T <= 16: 128 values, brought in once per workgroup through LDS); a workgroup takes a few output rows of one sample, stages the input rows they
// meet in LDS as fp32 with a zero border (no bounds logic in the loop), and its four waves walk the pixels of a row in
// turn: per pixel T broadcast reads from LDS, 8 T multiply-adds and one 16-byte (bf16) or two 16-byte (fp32) stores.
#include "conv_gather.h"

namespace gc {
namespace {

constexpr int kBorder = 3;            // tap offsets within [-3, 3]
constexpr int kRowsPerWg = 4;
constexpr int kMaxRowFloats = 1024;   // floats of one bordered input row set in LDS: (kRowsPerWg + 6) rows x (IW + 6)
constexpr int kLaneCh = 8, kBlockCh = 64 * kLaneCh;
constexpr int kSpreadTaps = 16;
constexpr int kWStride = kBlockCh + 4;

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void conv_spread_kernel(const GatherParams p, int row_groups) {
  __shared__ float gin[kMaxRowFloats];
  __shared__ __attribute__((aligned(16))) float wl[kSpreadTaps * kWStride];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = (int)blockIdx.x / row_groups, oh0 = ((int)blockIdx.x - n * row_groups) * kRowsPerWg;
  const int oc = (int)blockIdx.y * kBlockCh + lane * kLaneCh;
  const bool ch_ok = oc < p.OC;                       // (OC is a multiple of 8)
  const int T = p.tt.cls_tap_start[1];
  const int MW = p.IW + 2 * kBorder, MR = kRowsPerWg + 2 * kBorder;

  // ---- the input rows oh0 - 3 .. oh0 + kRowsPerWg + 2, columns -3 .. IW + 2 (zeros outside the map)
  const TIn* in = reinterpret_cast<const TIn*>(p.in);
  for (int e = t; e < MR * MW; e += kThreads) {
    const int mr = e / MW, ih = oh0 - kBorder + mr, iw = e - mr * MW - kBorder;
    float v = 0.f;
    if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) {
      const TIn raw = in[(((size_t)n * p.IH + ih) * p.IW + iw) * p.in_cstride + p.in_coff];
      // (an fp32 input is rounded to bf16 like the operands of every other kernel of this mode)
      v = sizeof(TIn) == 2 ? bf16_bits_to_float((uint16_t)raw) : bf16_bits_to_float(float_to_bf16_bits((float)raw));
    }
    gin[e] = v;
  }

  // ---- the weights of this workgroup's channel block: read tap-fastest (the 16 taps of a channel share one or two cache
  // lines of the packed image), parked in LDS as [tap][channel] (row stride 516 floats: a wave's 16 taps x 4 channels fall
  // into 64 different banks), then each lane takes the 16 x 8 values of its channels into registers
  const int oc_base = (int)blockIdx.y * kBlockCh;
  {
    const int tap = t & (kSpreadTaps - 1);             // (a thread keeps its tap: 256 is a multiple of 16)
    const int wt = tap < T ? (int)p.tt.wtap[tap] : -1;
    for (int e = t; e < kSpreadTaps * kBlockCh; e += kThreads) {
      const int o = e >> 4;
      float v = 0.f;
      if (wt >= 0 && oc_base + o < p.OC) v = bf16_bits_to_float(p.wp[(size_t)(oc_base + o) * p.row_pitch + (size_t)wt * p.ICp]);
      wl[tap * kWStride + o] = v;
    }
  }
  int toff[kSpreadTaps];
#pragma unroll
  for (int e = 0; e < kSpreadTaps; e++) toff[e] = e < T ? ((int)p.tt.dh[e] + kBorder) * MW + (int)p.tt.dw[e] + kBorder : 0;
  __syncthreads();
  float w[kSpreadTaps][kLaneCh];
#pragma unroll
  for (int e = 0; e < kSpreadTaps; e++) {
    const f32x4 lo = *reinterpret_cast<const f32x4*>(wl + e * kWStride + lane * kLaneCh);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(wl + e * kWStride + lane * kLaneCh + 4);
#pragma unroll
    for (int j = 0; j < 4; j++) { w[e][j] = lo[j]; w[e][4 + j] = hi[j]; }
  }
  float bias[kLaneCh];
#pragma unroll
  for (int j = 0; j < kLaneCh; j++) bias[j] = (p.bias != nullptr && ch_ok) ? p.bias[oc + j] : 0.f;
  const float sl = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
  if (!ch_ok) return;

  for (int r = 0; r < kRowsPerWg; r++) {
    const int oh = oh0 + r;
    if (oh >= p.OH) break;
    const float* grow = gin + r * MW;
    for (int ow = wave; ow < p.OW; ow += 4) {
      float v[kLaneCh];
#pragma unroll
      for (int j = 0; j < kLaneCh; j++) v[j] = 0.f;
#pragma unroll
      for (int e = 0; e < kSpreadTaps; e++) {
        const float g = grow[toff[e] + ow];
#pragma unroll
        for (int j = 0; j < kLaneCh; j++) v[j] += g * w[e][j];
      }
#pragma unroll
      for (int j = 0; j < kLaneCh; j++) {
        const float u = v[j] + bias[j];
        v[j] = fmaxf(u, u * sl);      // none / ReLU / LeakyReLU with a slope in [0, 1]
      }
      const size_t o = (((size_t)n * p.OH + oh) * p.OW + ow) * p.out_cstride + p.out_coff + oc;
      if (sizeof(TOut) == 4) {
        float* q = reinterpret_cast<float*>(p.out) + o;
        *reinterpret_cast<f32x4*>(q) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(q + 4) = f32x4{v[4], v[5], v[6], v[7]};
        if (p.out_bf16 != nullptr) {
          const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
          *reinterpret_cast<u32x4*>(p.out_bf16 + o) = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      } else {
        const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + o) = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
  }
}

}  // namespace

// One input channel, one class of at most 16 taps with offsets within [-3, 3], stride-1 gather, bf16 precision mode, at
// least 64 output channels in eights with 16-byte aligned pixels, plain epilogue (bias, none / ReLU / LeakyReLU), an input
// row set that fits the LDS table.
bool plan_spread(const GatherParams& p, const GatherLaunch& l) {
  static const bool off = getenv("GC_NO_SPREAD") != nullptr;
  if (off || l.x3 || p.IC != 1 || p.a != 1 || p.b != 1 || p.tt.n_classes != 1 || p.OC % 8 != 0 || p.OC < 64) return false;
  if (p.IH == 1 && p.IW == 1) return false;          // (one value per sample: the outer-product kernel)
  const int T = p.tt.cls_tap_start[1];
  if (p.tt.cls_tap_start[0] != 0 || T < 1 || T > kSpreadTaps) return false;
  for (int e = 0; e < T; e++)
    if (p.tt.dh[e] < -kBorder || p.tt.dh[e] > kBorder || p.tt.dw[e] < -kBorder || p.tt.dw[e] > kBorder) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.stat_rows != nullptr || p.mask != nullptr) return false;
  if (p.pre_scale != nullptr || p.post_scale != nullptr) return false;
  if (!(p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || (p.act == GC_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f))) return false;
  if ((reinterpret_cast<uintptr_t>(p.out) & 15u) != 0 || p.out_cstride % 8 != 0 || p.out_coff % 8 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.out_bf16) & 15u) != 0) return false;
  if ((kRowsPerWg + 2 * kBorder) * (p.IW + 2 * kBorder) > kMaxRowFloats) return false;
  const long groups = (long)p.N * ((p.OH + kRowsPerWg - 1) / kRowsPerWg);
  return groups > 0 && groups < (1l << 31) && (p.OC + kBlockCh - 1) / kBlockCh <= 65535;
}

int launch_spread(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  if (p.N == 0) return GC_OK;
  const int row_groups = (p.OH + kRowsPerWg - 1) / kRowsPerWg;
  const dim3 grid((unsigned)(p.N * row_groups), (unsigned)((p.OC + kBlockCh - 1) / kBlockCh)), block(kThreads);
  ProfScope ps("gc::conv_spread_kernel", l.flops, stream);
  if (l.in_bf16) {
    if (out_f32) hipLaunchKernelGGL((conv_spread_kernel<uint16_t, float>), grid, block, 0, stream, p, row_groups);
    else hipLaunchKernelGGL((conv_spread_kernel<uint16_t, uint16_t>), grid, block, 0, stream, p, row_groups);
  } else {
    if (out_f32) hipLaunchKernelGGL((conv_spread_kernel<float, float>), grid, block, 0, stream, p, row_groups);
    else hipLaunchKernelGGL((conv_spread_kernel<float, uint16_t>), grid, block, 0, stream, p, row_groups);
  }
  return check_hip(hipGetLastError(), "conv_spread launch");
}

}  // namespace gc
