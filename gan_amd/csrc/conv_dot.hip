// A convolution whose window is its whole (unpadded) input and that has at most four output channels: the critics'
// score layer, Conv2d(512 -> 1, kernel 4) on 4 x 4 maps (reference BasicGANs/models.py:77, last block).  Per sample it
// is one dot product of the NHWC input (which lies in memory in the packed weight row's own (tap, channel) order) with
// each weight row: a GEMV over 16 KB per sample, memory-bound.  The tiled kernels run it as ONE 256-pixel tile of 128
// reduction steps (or split the reduction with atomic adds); here a workgroup owns a sample, reads its input once with
// 16-byte loads, and sums in a fixed order -- bit-wise reproducible, a few microseconds.
#include "conv_gather.h"

namespace gc {
namespace {

template <typename TIn, typename TOut>
__global__ __launch_bounds__(kThreads) void conv_dot_kernel(const GatherParams p, int K) {
  __shared__ float red[kThreads / 64][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x;                 // a workgroup owns a sample, its waves interleave over the reduction
  const TIn* x = reinterpret_cast<const TIn*>(p.in) + (size_t)n * K;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k = (int)threadIdx.x * 8; k < K; k += kThreads * 8) {
    float xv[8];
    if (sizeof(TIn) == 2) {
      const u32x4 q = *reinterpret_cast<const u32x4*>(x + k);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        xv[2 * i] = __builtin_bit_cast(float, q[i] << 16);
        xv[2 * i + 1] = __builtin_bit_cast(float, q[i] & 0xffff0000u);
      }
    } else {      // fp32 input: rounded to bf16 like the operands of every other kernel of this precision mode
      const f32x4 a = *reinterpret_cast<const f32x4*>(x + k), b = *reinterpret_cast<const f32x4*>(x + k + 4);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        xv[i] = bf16_bits_to_float(float_to_bf16_bits(a[i]));
        xv[4 + i] = bf16_bits_to_float(float_to_bf16_bits(b[i]));
      }
    }
#pragma unroll
    for (int oc = 0; oc < 4; oc++) {
      if (oc >= p.OC) break;
      const u32x4 w = *reinterpret_cast<const u32x4*>(p.wp + (size_t)oc * p.row_pitch + k);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        acc[oc] += xv[2 * i] * __builtin_bit_cast(float, w[i] << 16);
        acc[oc] += xv[2 * i + 1] * __builtin_bit_cast(float, w[i] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int oc = 0; oc < 4; oc++) {
    if (oc >= p.OC) break;
    float s = acc[oc];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) red[wave][oc] = s;
  }
  __syncthreads();
#pragma unroll
  for (int oc = 0; oc < 4; oc++) {
    if (oc >= p.OC) break;
    if (threadIdx.x == 0) {
      float s = red[0][oc];
      for (int w = 1; w < kThreads / 64; w++) s += red[w][oc];      // fixed order
      if (p.bias != nullptr) s += p.bias[oc];
      s = apply_act(s, p.act, p.slope);
      const size_t o = (size_t)n * p.out_cstride + p.out_coff + oc;
      if (sizeof(TOut) == 4) reinterpret_cast<float*>(p.out)[o] = s;
      else reinterpret_cast<uint16_t*>(p.out)[o] = float_to_bf16_bits(s);
    }
  }
}

}  // namespace

// Forward pass with one output pixel per sample whose window is the whole dense input, channels in multiples of 64
// (the packed row then has the input's own order), at most four output channels, plain epilogue (bias, activation).
bool plan_dot(const GatherParams& p, const GatherLaunch& l) {
  if (l.x3 || l.mode != kVecLoads || p.OC > 4 || p.OH != 1 || p.OW != 1 || p.tt.n_classes != 1) return false;
  if (p.IC % kBK != 0 || p.ICp != p.IC || p.in_cstride != p.IC || p.in_coff != 0 || p.T != p.IH * p.IW || p.T > kMaxTaps) return false;
  if (p.row_pitch != p.T * p.IC || (reinterpret_cast<uintptr_t>(p.in) & 15u) != 0) return false;
  for (int e = 0; e < p.T; e++)
    if (p.tt.dh[e] != e / p.IW || p.tt.dw[e] != e % p.IW || p.tt.wtap[e] != e) return false;
  return p.ch_sum == nullptr && p.out_bf16 == nullptr && p.mask == nullptr;
}

int launch_dot(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  if (p.N == 0) return GC_OK;
  const int K = p.T * p.IC;
  const dim3 grid((unsigned)p.N), block(kThreads);
  ProfScope ps("gc::conv_dot_kernel", l.flops, stream);
  if (l.in_bf16) {
    if (out_f32) hipLaunchKernelGGL((conv_dot_kernel<uint16_t, float>), grid, block, 0, stream, p, K);
    else hipLaunchKernelGGL((conv_dot_kernel<uint16_t, uint16_t>), grid, block, 0, stream, p, K);
  } else {
    if (out_f32) hipLaunchKernelGGL((conv_dot_kernel<float, float>), grid, block, 0, stream, p, K);
    else hipLaunchKernelGGL((conv_dot_kernel<float, uint16_t>), grid, block, 0, stream, p, K);
  }
  return check_hip(hipGetLastError(), "conv_dot launch");
}

}  // namespace gc
