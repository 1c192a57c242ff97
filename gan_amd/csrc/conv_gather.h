// Gather-GEMM convolution kernel (forward and data-gradient passes) -- parameter block shared by host and device.
#pragma once
#include "common.h"

namespace gc {

// One entry per (output-pixel class, filter tap).  For the forward pass there is one class holding all taps;
// for the data gradient of a stride-b conv the output pixels split into b*b parity classes, each of which sees
// only the taps that land on it, so no zero-inserted work is done.
struct TapTable {
  int32_t n_classes;
  int32_t cls_tile_start[kMaxClasses + 1];  // first pixel-tile index of each class (prefix sums)
  int32_t cls_tap_start[kMaxClasses + 1];   // first entry of each class in dh/dw/wtap
  int8_t cls_ph[kMaxClasses], cls_pw[kMaxClasses];    // output pixel of class-local (oh', ow') is (oh'*b + ph, ow'*b + pw)
  int16_t cls_oh[kMaxClasses], cls_ow[kMaxClasses];   // class extent
  FastDiv cls_div_ow[kMaxClasses], cls_div_oh[kMaxClasses];   // division by the class extents (LDS-DMA kernel)
  int8_t dh[kMaxTaps], dw[kMaxTaps];        // input row / column of a tap: oh'*a + dh, ow'*a + dw
  uint8_t wtap[kMaxTaps];                   // index r*kw + s into the packed weight
};

struct GatherParams {
  const void* in;      // [N, IH, IW, in_cstride]
  const uint16_t* wp;  // packed bf16 [planes][OC][row_pitch]; a row is [tap][ICp] (+ zero tail up to row_pitch)
  void* out;           // [N, OH, OW, out_cstride]
  const float* bias;   // [OC] or null
  float* ch_sum;       // [OC] or null
  float* ch_sumsq;     // [OC] or null
  float* stat_rows;    // null, or [stat_cap][OC][2] row partials (sum, sum of squares); stat_written: host int, rows written
  int32_t stat_cap;
  int32_t* stat_written;
  void* split_ws;      // null, or scratch memory the kernels may use to split long reductions of small outputs over workgroups
  size_t split_ws_bytes;
  uint16_t* out_bf16;  // null, or a bf16 copy of the fp32 output (same indexing)
  const void* mask;    // null, or a tensor indexed like `out` (same element type): out *= mask > 0 ? 1 : slope
  const float* pre_scale;   // null, or [N][OC]: the accumulator of (sample n, channel oc) is multiplied by it in front of the bias
  const float* post_scale;  // null, or [N][OC]: the activated value is multiplied by it before it is stored
  int32_t N, IH, IW, IC, ICp, OH, OW, OC, T;
  int32_t row_pitch;   // elements per packed weight row
  int32_t a, b;        // input-coordinate multiplier, output class stride
  int32_t in_cstride, in_coff, out_cstride, out_coff;
  int32_t act;
  float slope;
  float mask_slope;    // the factor applied where mask <= 0
  int32_t out_vec;     // 1: 16-byte (fp32) / 8-byte (bf16) stores of 4 channels are aligned
  int32_t out_vec8;    // 1: 16-byte stores of 8 bf16 channels are aligned as well, and so are 16-byte loads of the bias
  int32_t out_f32;     // element type of `out`: 1 fp32, 0 bf16 (independent of the input's type)
  int32_t icp_log2;    // log2(ICp) in the thin-input mode (reduction over (tap, channel) jointly, 64 per step)
  int32_t ksplit;      // > 1: the reduction is split over ksplit workgroups that add into a zeroed fp32 output
  int32_t n_pix_tiles, n_oc_tiles;
  int32_t cls_interleave;  // 1: pixel tiles are ordered (tile, class) so that the classes sharing input rows run side by side
  int64_t plane_stride;  // elements between the hi and lo planes of wp
  int64_t in_bytes, w_bytes;  // extents of the input tensor and the packed weights (LDS-DMA path: buffer ranges)
  int32_t a_bias;        // LDS-DMA path: bytes by which the activation buffer starts before `in` (most negative tap offset)
  // few-output-channel kernel (conv_scatter): patch geometry
  int32_t sc_dh_min, sc_dw_min, sc_ph, sc_pw, sc_tiles_h, sc_tiles_w, sc_th;
  FastDiv sc_div_oc;
  TapTable tt;
};

enum GatherMode { kVecLoads = 0, kElemLoads = 1, kThinInput = 2 };

struct GatherLaunch {
  int to, tp;          // tile: output channels x pixels
  bool in_bf16, x3;
  int mode;            // GatherMode
  bool big_tile_ok;    // the caller allows the 256-pixel tile of the LDS-DMA path
  double flops;        // algorithmic FLOPs of this launch (for the optional event timing)
};

// Packed-weight geometry for `ic` channels on the reduction side and `taps` filter taps.
struct PackGeom {
  int icp, row_pitch, thin;
};
PackGeom pack_geom(int ic, int taps, bool allow_thin);

// Chooses the tile for a problem; fills n_pix_tiles / n_oc_tiles / ksplit and the class tile prefix in p.tt.
void plan_gather(GatherParams& p, GatherLaunch& l);
int launch_gather(const GatherParams& p, const GatherLaunch& l, hipStream_t stream);
// Few output channels (at most 64 (tap, channel) pairs), stride-1 gather: fills the patch geometry, false if unsuitable.
bool plan_scatter(GatherParams& p, const GatherLaunch& l);
int launch_scatter(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
// Image layer (<= 4 fp32 input channels, 4x4 stride-2 forward): patch staged once, see conv_image.hip.
bool plan_dot(const GatherParams& p, const GatherLaunch& l);
int launch_dot(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
bool plan_image(const GatherParams& p, const GatherLaunch& l);
bool image_applies_mask(const GatherParams& p);      // launch_image multiplies by the gradient mask itself
// 1 x 1 convolution of an image of at most 4 channels (StyleGAN's from-RGB layers), see conv_point.hip; leaves a gradient
// mask to the caller.
bool plan_point(const GatherParams& p, const GatherLaunch& l);
int launch_point(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
// One input value per sample (the data gradient of a critic's score layer): an outer product, see conv_outer.hip; applies
// the gradient mask itself.
bool plan_outer(const GatherParams& p, const GatherLaunch& l);
int launch_outer(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
// One input channel, many output channels, one class of up to 16 taps (the data gradient of a sliding-window score layer).
bool plan_spread(const GatherParams& p, const GatherLaunch& l);
int launch_spread(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
int launch_image(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
// Patch-resident kernel (stride-1 gathers with >= 4 taps per class on large images, see conv_patch.hip); plan_patch
// rewrites the tile bookkeeping in p when it accepts.
bool plan_patch(GatherParams& p, const GatherLaunch& l);
int launch_patch(GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);   // clears p.mask when it applied it
// Weights-resident persistent kernel for stride-1 gathers over at most 64 input channels on large images (conv_thin.hip);
// applies the gradient mask itself.
bool plan_thin(const GatherParams& p, const GatherLaunch& l);
int launch_thin(GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
// Persistent wave-specialised LDS-DMA kernel for mid-size layers (conv_gather_ws.hip): full channel tiles of 64 / 128,
// 128-pixel tiles, 16-byte aligned output rows, none / ReLU / LeakyReLU epilogues.
bool plan_ws(const GatherParams& p, const GatherLaunch& l);
int launch_ws(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);
// LDS-DMA fast path: bf16 input, vector-aligned channels, 64- or 128-channel tile, no reduction split.
int launch_gather_dma(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t stream);

}  // namespace gc
