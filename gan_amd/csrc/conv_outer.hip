// A gather whose input is ONE value per sample (one channel, one pixel): the data gradient of the critics' score layer,
// Conv2d(512 -> 1, kernel 4) on 4 x 4 maps (reference BasicGANs/models.py:77, last block; the WGAN-GP penalty runs it a
// second time under create_graph, BasicGANs/loss_functions.py:18-28).  Every output pixel sees at most one filter tap,
// so the result is the outer product  out[n][pixel][oc] = in[n] * w[oc][tap(pixel)]  (+ bias, activation, gradient
// mask): 8 KFLOP per sample against 16 KB written.  The tiled kernels run it as a GEMM with a reduction of length one
// padded to 64; here a thread owns one (pixel, 8 output channels) pair, keeps its eight weights in registers and walks
// a run of samples, one 16-byte store each.
#include <string.h>

#include "conv_gather.h"

namespace gc {
namespace {

constexpr int kSamplesPerThread = 4;
constexpr int kMaxPixels = 64;      // output pixels per sample

struct OuterTaps {
  uint32_t v[kMaxPixels / 4];       // byte q: the weight tap of output pixel q, 255 when no tap reaches it
};

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void conv_outer_kernel(const GatherParams p, const OuterTaps tab, uint32_t pairs, uint32_t oc8s) {
  const uint32_t pair = blockIdx.x * kThreads + threadIdx.x;
  const uint32_t pix = min(pair / oc8s, (uint32_t)(kMaxPixels - 1)), oc = (pair - pix * oc8s) * 8;
  // the filter tap that reaches this pixel (255: none), from the table the host worked out: one byte per pixel, handed
  // over as kernel arguments and spread through LDS (a thread cannot index kernel arguments by its own pixel)
  __shared__ uint32_t tap_words[kMaxPixels / 4];
#pragma unroll
  for (int k = 0; k < kMaxPixels / 4; k++)
    if (threadIdx.x == k) tap_words[k] = tab.v[k];
  __syncthreads();
  if (pair >= pairs) return;
  const int wt_byte = (int)((tap_words[pix >> 2] >> (8 * (pix & 3))) & 0xffu);
  const int wt = wt_byte == 255 ? -1 : wt_byte;
  float w[8], bias[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    w[j] = wt >= 0 ? bf16_bits_to_float(p.wp[(size_t)(oc + j) * p.row_pitch + (size_t)wt * p.ICp]) : 0.f;
    bias[j] = p.bias != nullptr ? p.bias[oc + j] : 0.f;
  }
  const float sl = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
  const TIn* in = reinterpret_cast<const TIn*>(p.in);
  const int n0 = (int)blockIdx.y * kSamplesPerThread, n1 = min(p.N, n0 + kSamplesPerThread);
  const size_t per_sample = (size_t)p.OH * p.OW * p.out_cstride;
  const size_t o0 = (size_t)pix * p.out_cstride + p.out_coff + oc;
  // the samples' values: all loads first (a sample past the end repeats the last one and is never stored), conversions
  // after them, so that the loads are in flight together
  TIn raw[kSamplesPerThread];
#pragma unroll
  for (int i = 0; i < kSamplesPerThread; i++) raw[i] = in[(size_t)min(n0 + i, p.N - 1) * p.in_cstride + p.in_coff];
  float xs[kSamplesPerThread];
#pragma unroll
  for (int i = 0; i < kSamplesPerThread; i++)      // fp32 input: rounded to bf16 like the operands of every other kernel of this mode
    xs[i] = sizeof(TIn) == 2 ? bf16_bits_to_float((uint16_t)raw[i]) : bf16_bits_to_float(float_to_bf16_bits((float)raw[i]));
  f32x4 m[kSamplesPerThread][2];
  if (p.mask != nullptr) {
#pragma unroll
    for (int i = 0; i < kSamplesPerThread; i++) {
      const size_t o = (size_t)min(n0 + i, p.N - 1) * per_sample + o0;
      m[i][0] = load4<TOut>(p.mask, o);
      m[i][1] = load4<TOut>(p.mask, o + 4);
    }
  }
#pragma unroll
  for (int i = 0; i < kSamplesPerThread; i++) {
    if (n0 + i >= n1) break;
    const size_t o = (size_t)(n0 + i) * per_sample + o0;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float u = 0.f + xs[i] * w[j] + bias[j];
      v[j] = fmaxf(u, u * sl);      // none / ReLU / LeakyReLU with a slope in [0, 1]
    }
    if (p.mask != nullptr) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = m[i][0][j] > 0.f ? v[j] : v[j] * p.mask_slope;
        v[4 + j] = m[i][1][j] > 0.f ? v[4 + j] : v[4 + j] * p.mask_slope;
      }
    }
    const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
    if (sizeof(TOut) == 4) {
      float* q = reinterpret_cast<float*>(p.out) + o;
      *reinterpret_cast<f32x4*>(q) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(q + 4) = f32x4{v[4], v[5], v[6], v[7]};
      if (p.out_bf16 != nullptr) *reinterpret_cast<u32x4*>(p.out_bf16 + o) = u32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
      *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + o) = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }
  }
}

}  // namespace

// One input value per sample (one channel on a 1 x 1 map), bf16 precision mode, output channels in eights with 16-byte
// aligned rows, no channel statistics, activation none / ReLU / LeakyReLU, any class stride (the data gradient of a window that covers its whole input comes
// with one class per pixel).  Every class's taps have distinct offsets, so at most one of them lands on the input pixel.
bool plan_outer(const GatherParams& p, const GatherLaunch& l) {
  if (l.x3 || p.IC != 1 || p.IH != 1 || p.IW != 1 || p.OC % 8 != 0 || p.b < 1) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.stat_rows != nullptr) return false;
  if (!(p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || (p.act == GC_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f))) return false;
  if ((reinterpret_cast<uintptr_t>(p.out) & 15u) != 0 || p.out_cstride % 8 != 0 || p.out_coff % 8 != 0) return false;
  if ((reinterpret_cast<uintptr_t>(p.out_bf16) & 15u) != 0 || (reinterpret_cast<uintptr_t>(p.mask) & 15u) != 0) return false;
  if (p.OH * p.OW > kMaxPixels || p.T > 254 || (p.N + kSamplesPerThread - 1) / kSamplesPerThread > 65535) return false;
  for (int c = 0; c < p.tt.n_classes; c++)
    for (int e = p.tt.cls_tap_start[c]; e < p.tt.cls_tap_start[c + 1]; e++)
      for (int f = e + 1; f < p.tt.cls_tap_start[c + 1]; f++)
        if (p.tt.dh[e] == p.tt.dh[f] && p.tt.dw[e] == p.tt.dw[f]) return false;
  return true;
}

int launch_outer(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  if (p.N == 0) return GC_OK;
  const uint32_t oc8s = (uint32_t)(p.OC / 8), pairs = (uint32_t)p.OH * p.OW * oc8s;
  const dim3 grid((pairs + kThreads - 1) / kThreads, (unsigned)((p.N + kSamplesPerThread - 1) / kSamplesPerThread)), block(kThreads);
  OuterTaps tab;
  memset(&tab, 0xff, sizeof(tab));
  for (int oh = 0; oh < p.OH; oh++)
    for (int ow = 0; ow < p.OW; ow++) {
      const int ohc = oh / p.b, owc = ow / p.b, q = oh * p.OW + ow;
      for (int c = 0; c < p.tt.n_classes; c++) {
        if (p.tt.cls_ph[c] != oh % p.b || p.tt.cls_pw[c] != ow % p.b) continue;
        for (int e = p.tt.cls_tap_start[c]; e < p.tt.cls_tap_start[c + 1]; e++)
          if (ohc * p.a + p.tt.dh[e] == 0 && owc * p.a + p.tt.dw[e] == 0)
            tab.v[q >> 2] = (tab.v[q >> 2] & ~(0xffu << (8 * (q & 3)))) | ((uint32_t)p.tt.wtap[e] << (8 * (q & 3)));
      }
    }
  ProfScope ps("gc::conv_outer_kernel", l.flops, stream);
  if (l.in_bf16) {
    if (out_f32) hipLaunchKernelGGL((conv_outer_kernel<uint16_t, float>), grid, block, 0, stream, p, tab, pairs, oc8s);
    else hipLaunchKernelGGL((conv_outer_kernel<uint16_t, uint16_t>), grid, block, 0, stream, p, tab, pairs, oc8s);
  } else {
    if (out_f32) hipLaunchKernelGGL((conv_outer_kernel<float, float>), grid, block, 0, stream, p, tab, pairs, oc8s);
    else hipLaunchKernelGGL((conv_outer_kernel<float, uint16_t>), grid, block, 0, stream, p, tab, pairs, oc8s);
  }
  return check_hip(hipGetLastError(), "conv_outer launch");
}

}  // namespace gc
