// Weight gradient of a convolution with ONE output channel and a small stride-1 filter (at most 4 x 4) over a wide input:
// the score layers of the critics (PatchGAN: Conv2d(512 -> 1, kernel 4, stride 1, padding 1) on 31 x 31 maps, reference
// Pix2Pix/models.py:238; DCGAN / WGAN-GP: the 4 x 4 window over the whole 4 x 4 map, BasicGANs/models.py:77).
//
//   gw[0][ci][r][s] = sum over (n, ho, wo) of gy[n][ho][wo] * x[n][ho - ph + r][wo - pw + s][ci]
//
// 2 * T FLOP per element of x: bound by reading x once.  The tiled kernels treat every tap as a GEMM of its own with 63 of 64
// output rows empty and read x once per tap.  Here the loop runs over the pixels of x: pixel (ih, iw) adds
// gy[ih + ph - r][iw + pw - s] * x[ih][iw][:] to tap (r, s).  The gradient map of the current sample waits in LDS as fp32
// with a zero border of three pixels (no bounds logic in the loop); wave r of a workgroup owns filter row r, a lane owns 8
// consecutive channels (16-byte loads of x, a burst of 16 pixels in flight), so the accumulators are 4 taps x 8 channels and
// the four waves write disjoint taps: no reduction inside the workgroup.  A split is a run of image rows (n, ih); the
// partial sums leave in the common layout [split][tap][1][Cin] and the common split reduction follows.
#include "conv_wgrad.h"

namespace gc {
namespace {

constexpr int kBorder = 3;           // filter extents up to 4
constexpr int kMaxMap = 4096;        // floats of the bordered gradient map in LDS
constexpr int kLaneCh = 8, kBlockCh = 64 * kLaneCh;
constexpr int kBurst = 16;

__global__ __launch_bounds__(kThreads) void conv_wgrad_score_kernel(const WgradParams p) {
  __shared__ float gmap[kMaxMap];
  const int t = threadIdx.x, lane = t & 63, r = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cblocks = (p.Cin + kBlockCh - 1) / kBlockCh;
  const int split = (int)blockIdx.x / cblocks, cb = (int)blockIdx.x - split * cblocks;
  const int ci = cb * kBlockCh + lane * kLaneCh;
  const bool active = r < p.kh && ci < p.Cin;
  const int MW = p.Wo + 2 * kBorder, MH = p.Ho + 2 * kBorder;
  const int u0 = split * p.k_per_split, u1 = min(p.N * p.H, u0 + p.k_per_split);
  const uint16_t* x = reinterpret_cast<const uint16_t*>(p.x);
  const uint16_t* gy = reinterpret_cast<const uint16_t*>(p.gy);

  float acc[4][kLaneCh];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int c = 0; c < kLaneCh; c++) acc[s][c] = 0.f;

  int cur_n = -1;
  for (int u = u0; u < u1; u++) {
    const int n = u / p.H, ih = u - n * p.H;
    if (n != cur_n) {                      // (the same for every thread of the workgroup)
      __syncthreads();                     // the previous sample's map has been read
      for (int e = t; e < MH * MW; e += kThreads) {
        const int mh = e / MW, hh = mh - kBorder, ww = e - mh * MW - kBorder;
        float v = 0.f;
        if ((unsigned)hh < (unsigned)p.Ho && (unsigned)ww < (unsigned)p.Wo)
          v = bf16_bits_to_float(gy[(((size_t)n * p.Ho + hh) * p.Wo + ww) * p.y_cstride + p.y_coff]);
        gmap[e] = v;
      }
      __syncthreads();
      cur_n = n;
    }
    const int ho = ih + p.ph - r;          // the gradient row this wave's filter row meets on image row ih
    if (!active || (unsigned)ho >= (unsigned)p.Ho) continue;
    const float* grow = gmap + (ho + kBorder) * MW + kBorder + p.pw;      // tap column s of pixel iw: grow[iw - s]
    const uint16_t* xrow = x + ((size_t)n * p.H + ih) * p.W * p.x_cstride + p.x_coff + ci;
    for (int iw0 = 0; iw0 < p.W; iw0 += kBurst) {
      u32x4 xv[kBurst];
#pragma unroll
      for (int j = 0; j < kBurst; j++) {
        xv[j] = u32x4{0u, 0u, 0u, 0u};
        if (iw0 + j < p.W) xv[j] = *reinterpret_cast<const u32x4*>(xrow + (size_t)(iw0 + j) * p.x_cstride);
      }
#pragma unroll
      for (int j = 0; j < kBurst; j++) {
        if (iw0 + j >= p.W) break;
        float xf[kLaneCh];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          xf[2 * q] = __builtin_bit_cast(float, xv[j][q] << 16);
          xf[2 * q + 1] = __builtin_bit_cast(float, xv[j][q] & 0xffff0000u);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
          const float g = grow[iw0 + j - s];
#pragma unroll
          for (int c = 0; c < kLaneCh; c++) acc[s][c] += g * xf[c];
        }
      }
    }
  }
  if (!active) return;
  const int T = p.kh * p.kw;
#pragma unroll
  for (int s = 0; s < 4; s++)
    if (s < p.kw) {
      float* dst = p.part + ((size_t)split * T + r * p.kw + s) * p.Cin + ci;
      *reinterpret_cast<f32x4*>(dst) = f32x4{acc[s][0], acc[s][1], acc[s][2], acc[s][3]};
      *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[s][4], acc[s][5], acc[s][6], acc[s][7]};
    }
}

}  // namespace

// Takes the problem when the geometry is the kernel's: one output channel, filter extents up to 4, stride 1, bf16 operands,
// at least 128 input channels in multiples of 8 with 16-byte aligned pixels, a gradient map that fits the LDS table, one
// (x, gy) pair.  A split becomes a run of image rows: k_per_split = rows per split.
bool plan_wgrad_score(WgradParams& p, const WgradLaunch& l, bool operands_aligned) {
  static const bool off = getenv("GC_NO_WGRAD_SCORE") != nullptr;
  if (off || !operands_aligned || !l.bf16 || l.x3 || p.x_f32 || p.Cout != 1 || p.kh > 4 || p.kw > 4 || p.sh != 1 || p.sw != 1) return false;
  if (p.ph >= p.kh || p.pw >= p.kw || p.Cin < 128 || p.Cin % 8 != 0 || p.x_cstride % 8 != 0 || p.x_coff % 8 != 0) return false;
  if ((long)(p.Ho + 2 * kBorder) * (p.Wo + 2 * kBorder) > kMaxMap || p.N <= 0) return false;
  const long rows = (long)p.N * p.H;
  if (rows >= (1l << 30)) return false;
  const int cblocks = (p.Cin + kBlockCh - 1) / kBlockCh;
  long want = (512 + cblocks - 1) / cblocks;           // about two workgroups per CU
  if (want > rows) want = rows;
  const long per = (rows + want - 1) / want;
  const long splits = (rows + per - 1) / per;
  const size_t bytes = (size_t)splits * p.kh * p.kw * p.Cin * sizeof(float) + (size_t)splits * sizeof(float) + 16;
  if (wgrad_rows_workspace_cap() != 0 && bytes > wgrad_rows_workspace_cap()) return false;      // a caller with a bounded workspace
  p.k_per_split = (int32_t)per;
  p.splits = (int32_t)splits;
  p.score = 1;
  return true;
}

int launch_wgrad_score(const WgradParams& p, const WgradLaunch& l, hipStream_t stream) {
  if (p.splits <= 0 || p.N <= 0) return GC_OK;
  ProfScope ps("gc::conv_wgrad_score_kernel", l.flops, stream);
  const int cblocks = (p.Cin + kBlockCh - 1) / kBlockCh;
  hipLaunchKernelGGL(conv_wgrad_score_kernel, dim3((unsigned)(p.splits * cblocks)), dim3(kThreads), 0, stream, p);
  return check_hip(hipGetLastError(), "conv_wgrad_score launch");
}

}  // namespace gc
