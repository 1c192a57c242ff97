// Weight-gradient kernel: parameter block shared by host and device.
#pragma once
#include "common.h"

namespace gc {

constexpr int kMaxWgradSegments = 4;

struct WgradParams {
  const void* x;    // [N, H, W, x_cstride]
  const void* gy;   // [N, Ho, Wo, y_cstride]
  float* part;      // split-K partial sums [S][T][Cout][Cin] fp32
  float* bias_part; // null, or [S][Cout] fp32: per split the sum of gy over its pixels (LDS-DMA and row-streaming kernels)
  int32_t N, H, W, Cin, Ho, Wo, Cout, kh, kw, sh, sw, ph, pw;
  int32_t x_cstride, x_coff, y_cstride, y_coff;
  int32_t K;             // N*Ho*Wo: length of the reduction
  int32_t k_per_split;   // multiple of the reduction step
  int32_t n_co_tiles, n_ci_tiles, splits;
  int32_t y_elem;        // register-staged kernel, vector instantiation: the gy side still needs element loads (odd Cout)
  int32_t x_f32;         // 1: x is fp32 while gy is bf16 (the image-layer kernel rounds it in registers; nothing else takes the mix)
  int32_t image;         // 1: geometry of the image-layer kernel (conv_wgrad_image.hip); splits start on its 128-pixel rectangles
  int32_t score;         // 1: planned for the one-output-channel kernel (conv_wgrad_score.hip): a split is a run of k_per_split image rows
  int32_t thin;          // 1: Cin <= 32; the tile's columns run over (tap, channel) jointly, channel padded to ICp
  int32_t ICp, icp_log2, ncols;   // thin mode: padded channels per tap, its log2, T * ICp
  int32_t n_tap_groups, tpt, cw_log2;   // LDS-DMA kernel: a 128-column tile holds tpt taps of (1 << cw_log2) channels
  FastDiv div_wo, div_ho;
  int64_t x_bytes, y_bytes;   // extents of x and gy (buffer ranges of the LDS-DMA kernel)
  // Several (x, gy) pairs of one layer reduced into one gradient (the passes of a training phase that all add to the
  // same weight gradient): n_seg > 1, segment g owns the splits seg_split0[g] .. seg_split0[g + 1] - 1 and K = seg_K[g].
  int32_t n_seg;
  // Row-streaming 3x3 kernel (conv_wgrad_rows.hip): rows = 1; the reduction is dealt out in column strips of 32 pixels
  // (strips_w per image row, total_strips over all samples of all segments, strips_per_split per workgroup) and
  // seg_split0[g] is the first STRIP of segment g.
  int32_t rows, strips_w, total_strips, strips_per_split;
  int32_t seg_K[kMaxWgradSegments];
  int32_t seg_split0[kMaxWgradSegments];
  const void* seg_x[kMaxWgradSegments];
  const void* seg_gy[kMaxWgradSegments];
  int64_t seg_x_bytes[kMaxWgradSegments], seg_y_bytes[kMaxWgradSegments];
};

// The tensors and the pixel range a workgroup of split `split` reduces over.
struct WgradRange {
  const void* x;
  const void* gy;
  int64_t x_bytes, y_bytes;
  int kb0, kend;
};
__device__ __forceinline__ WgradRange wgrad_range(const WgradParams& p, int split) {
  WgradRange r{p.x, p.gy, p.x_bytes, p.y_bytes, split * p.k_per_split, 0};
  int K = p.K;
  if (p.n_seg > 1) {
#pragma unroll
    for (int g = 0; g < kMaxWgradSegments; g++)      // (static indices: the argument block is read at fixed offsets)
      if (g < p.n_seg && split >= p.seg_split0[g]) {
        r.x = p.seg_x[g]; r.gy = p.seg_gy[g]; r.x_bytes = p.seg_x_bytes[g]; r.y_bytes = p.seg_y_bytes[g];
        r.kb0 = (split - p.seg_split0[g]) * p.k_per_split;
        K = p.seg_K[g];
      }
  }
  r.kend = min(K, r.kb0 + p.k_per_split);
  return r;
}

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
  const float* bias_part;   // null, or [splits][Cout]: the extra workgroups of the launch add them in split order into gbias
  float* gbias;
  int32_t main_blocks;      // workgroups in front of the bias ones
};

void plan_wgrad(WgradParams& p, WgradLaunch& l, bool allow_dma);
int launch_wgrad(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
int launch_wgrad_dma(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
bool wgrad_image_geometry(const WgradParams& p);
bool wgrad_image_applies(const WgradParams& p, const WgradLaunch& l);
int launch_wgrad_image(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
bool plan_wgrad_rows(WgradParams& p, const WgradLaunch& l, int n_seg, const int32_t* Ns);
size_t wgrad_rows_workspace_cap();    // 0, or the calling thread's bound on the partial sums of the row-streaming plan
int launch_wgrad_rows(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
bool plan_wgrad_score(WgradParams& p, const WgradLaunch& l, bool operands_aligned);
int launch_wgrad_score(const WgradParams& p, const WgradLaunch& l, hipStream_t stream);
int launch_wgrad_reduce(const WgradReduceParams& p, hipStream_t stream);
int launch_colsum(const void* gy, float* gb, int K, int C, int cstride, int coff, bool bf16, hipStream_t stream);

}  // namespace gc
