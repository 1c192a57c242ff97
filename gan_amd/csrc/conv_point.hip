// 1 x 1 convolution of an image with at most four channels: the from-RGB layers of the StyleGAN discriminator
// (reference StyleGAN/model.py:453-514: EqualizedConv2d(3, C, 1) on 256 x 256 and 128 x 128 images).  12 FLOP per output
// element against 2 bytes written: bound by the store of the output (268 MB at batch 32, 64 channels, 256 x 256).  The
// thin-input tile kernel pads the reduction of 3 to 64 and fetches every input element with a load of its own; here a
// thread keeps the weights of 8 output channels (at most 32 values) in registers and walks pixels: one 16-byte store
// per pixel, the 8 lanes that share a pixel read the same 4 to 16 input bytes.
#include "conv_gather.h"

namespace gc {
namespace {

constexpr int kPixPerStep = 4;      // pixels a thread has in flight

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void conv_point_kernel(const GatherParams p, uint32_t pixels, int oc8s_log2) {
  const int t = threadIdx.x;
  const int oc = (t & ((1 << oc8s_log2) - 1)) * 8;
  const int lanes_pix = kThreads >> oc8s_log2;            // pixels a workgroup covers per pass
  float w[8][4], bias[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
#pragma unroll
    for (int c = 0; c < 4; c++) w[j][c] = c < p.IC ? bf16_bits_to_float(p.wp[(size_t)(oc + j) * p.row_pitch + c]) : 0.f;
    bias[j] = p.bias != nullptr ? p.bias[oc + j] : 0.f;
  }
  const float sl = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);
  const TIn* in = reinterpret_cast<const TIn*>(p.in);
  const uint32_t stride = gridDim.x * (uint32_t)lanes_pix;
  for (uint32_t q0 = blockIdx.x * (uint32_t)lanes_pix + (uint32_t)(t >> oc8s_log2); q0 < pixels; q0 += stride * kPixPerStep) {
    // all loads of the step first, conversions after them
    TIn raw[kPixPerStep][4];
#pragma unroll
    for (int i = 0; i < kPixPerStep; i++) {
      const uint32_t q = min(q0 + (uint32_t)i * stride, pixels - 1);
      const TIn* src = in + (size_t)q * p.in_cstride + p.in_coff;
#pragma unroll
      for (int c = 0; c < 4; c++) raw[i][c] = src[c < p.IC ? c : 0];
    }
#pragma unroll
    for (int i = 0; i < kPixPerStep; i++) {
      const uint32_t q = q0 + (uint32_t)i * stride;
      if (q >= pixels) break;
      float x[4];
#pragma unroll
      for (int c = 0; c < 4; c++)      // fp32 input: rounded to bf16 like the operands of every other kernel of this mode
        x[c] = c >= p.IC ? 0.f : sizeof(TIn) == 2 ? bf16_bits_to_float((uint16_t)raw[i][c]) : bf16_bits_to_float(float_to_bf16_bits((float)raw[i][c]));
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float u = x[0] * w[j][0];
#pragma unroll
        for (int c = 1; c < 4; c++) u += x[c] * w[j][c];       // (values and weights of channels past IC are zero)
        u += bias[j];
        v[j] = fmaxf(u, u * sl);      // none / ReLU / LeakyReLU with a slope in [0, 1]
      }
      const size_t o = (size_t)q * p.out_cstride + p.out_coff + oc;
      if (sizeof(TOut) == 4) {
        float* dst = reinterpret_cast<float*>(p.out) + o;
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{v[4], v[5], v[6], v[7]};
      } else {
        const u32x2 lo = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]}), hi = cvt_bf16x4(f32x4{v[4], v[5], v[6], v[7]});
        *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + o) = u32x4{lo[0], lo[1], hi[0], hi[1]};
      }
    }
  }
}

}  // namespace

// A 1 x 1 / stride 1 forward pass of at most 4 input channels on large images, bf16 precision mode, 8 .. 256 output channels
// (a power of two of eights per workgroup), 16-byte aligned output rows, plain epilogue (bias, none / ReLU / LeakyReLU).
bool plan_point(const GatherParams& p, const GatherLaunch& l) {
  if (l.x3 || p.IC > 4 || p.T != 1 || p.a != 1 || p.b != 1 || p.tt.n_classes != 1 || p.tt.dh[0] != 0 || p.tt.dw[0] != 0) return false;
  if (p.IH != p.OH || p.IW != p.OW || (long)p.N * p.OH * p.OW < 65536 || (long)p.N * p.OH * p.OW >= (1l << 31)) return false;
  const int oc8s = p.OC / 8;
  if (p.OC % 8 != 0 || oc8s > 32 || (oc8s & (oc8s - 1)) != 0) return false;
  if (p.ch_sum != nullptr || p.ch_sumsq != nullptr || p.stat_rows != nullptr || p.out_bf16 != nullptr) return false;
  if (!(p.act == GC_ACT_NONE || p.act == GC_ACT_RELU || (p.act == GC_ACT_LRELU && p.slope >= 0.f && p.slope <= 1.f))) return false;
  return (reinterpret_cast<uintptr_t>(p.out) & 15u) == 0 && p.out_cstride % 8 == 0 && p.out_coff % 8 == 0;
}

int launch_point(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  const uint32_t pixels = (uint32_t)((long)p.N * p.OH * p.OW);
  int oc8s_log2 = 0;
  while ((8 << oc8s_log2) < p.OC) oc8s_log2++;
  const uint32_t per_pass = (uint32_t)(kThreads >> oc8s_log2) * kPixPerStep;
  uint32_t blocks = (pixels + per_pass - 1) / per_pass;
  if (blocks > 256 * 16) blocks = 256 * 16;
  const dim3 grid(blocks), block(kThreads);
  ProfScope ps("gc::conv_point_kernel", l.flops, stream);
  if (l.in_bf16) {
    if (out_f32) hipLaunchKernelGGL((conv_point_kernel<uint16_t, float>), grid, block, 0, stream, p, pixels, oc8s_log2);
    else hipLaunchKernelGGL((conv_point_kernel<uint16_t, uint16_t>), grid, block, 0, stream, p, pixels, oc8s_log2);
  } else {
    if (out_f32) hipLaunchKernelGGL((conv_point_kernel<float, float>), grid, block, 0, stream, p, pixels, oc8s_log2);
    else hipLaunchKernelGGL((conv_point_kernel<float, uint16_t>), grid, block, 0, stream, p, pixels, oc8s_log2);
  }
  return check_hip(hipGetLastError(), "conv_point launch");
}

}  // namespace gc
