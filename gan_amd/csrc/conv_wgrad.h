// Weight-gradient kernel: parameter block shared by host and device.
#pragma once
#include "common.h"

namespace gc {

struct WgradParams {
  const void* x;    // [N, H, W, x_cstride]
  const void* gy;   // [N, Ho, Wo, y_cstride]
  float* part;      // split-K partial sums [S][T][Cout][Cin] fp32
  int32_t N, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw;
  int32_t x_cstride, x_coff, y_cstride, y_coff;
  int32_t K;             // N*Ho*Wo: length of the reduction
  int32_t k_per_split;   // multiple of the reduction step
  int32_t n_co_tiles, n_ci_tiles, splits;
  int32_t y_elem;        // register-staged kernel, vector instantiation: the gy side still needs element loads (odd Cout)
  int32_t thin;          // 1: Cin <= 32; the tile's columns run over (tap, channel) jointly, channel padded to ICp
  int32_t ICp, icp_log2, ncols;   // thin mode: padded channels per tap, its log2, T * ICp
  int32_t n_tap_groups, tpt, cw_log2;   // LDS-DMA kernel: a 128-column tile holds tpt taps of (1 << cw_log2) channels
  FastDiv div_wo, div_ho;
  int64_t x_bytes, y_bytes;   // extents of x and gy (buffer ranges of the LDS-DMA kernel)
};

struct WgradLaunch {
  int to, tc;
  bool bf16, x3, aligned;
  bool dma;   // planned for the LDS-DMA kernel (bf16 operands, channel counts and strides multiples of 8, Cin > 32)
  double flops;
};

struct WgradReduceParams {
  const float* part;
  float* gw;
  int64_t s_co, s_ci, s_r, s_s;   // element strides of gw
  int32_t Cout, Cin, kh, kw, splits;
  float alpha;
  int32_t accumulate;
  int32_t thin, ICp;     // layout of part: [split][tap][Cout][Cin], or thin: [split][Cout][tap][ICp]
};

void plan_wgrad(WgradParams& p, WgradLaunch& l, bool allow_dma);
int launch_wgrad(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
int launch_wgrad_dma(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
int launch_wgrad_reduce(const WgradReduceParams& p, hipStream_t stream);
int launch_colsum(const void* gy, float* gb, int K, int C, int cstride, int coff, bool bf16, hipStream_t stream);

}  // namespace gc
