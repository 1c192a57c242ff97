// Weight gradient of the critics' and the U-Net's first layer: Conv2d(<= 4 channels -> up to 64, kernel 4, stride 2,
// padding 1) on images whose output width is a multiple of 32 (reference BasicGANs/models.py:52-56,77: 64-pixel-wide
// images; Pix2Pix/models.py:8-20: 256-pixel-wide ones), bf16 operands.  3 MFLOP per sample against 270 KB of
// gy and image: memory-bound, so gy and the image are each read once and the per-pixel instruction count is kept low
// (the general thin-input kernel gathers every (tap, channel) element of its x tile with a 2-byte load of its own).
//
// A workgroup walks the rectangles (4 output rows x 32 columns = 128 values of the reduction index; units are numbered
// sample-major, then rows of rectangles, then columns, and a split is a run of units) of its split.  Per
// rectangle it stages, as the forward kernel of this layer does, the 10 x 66 input patch as channel-padded bf16 -- the 4
// taps x 4 channels of one filter row are then 32 contiguous bytes per output pixel, so the x operand of an MFMA whose
// reduction runs over 32 pixels is two transposing LDS reads straight from the patch, no im2col tile -- and the 128 x 64
// gy tile (16-byte loads, rows swizzled in 32-byte segments for the transposing reads).  Wave w owns output row w of the
// rectangle: 16 MFMAs (4 blocks of 16 output channels x 4 filter rows) per rectangle into 64 accumulator registers that
// live across the whole split.  Loads of rectangle i + 1 are in flight while rectangle i is multiplied (two LDS
// buffers).  At the end the four waves' tiles are added through LDS and leave as one partial tile in the layout of the
// thin-input kernel ([split][Cout][tap][4 channels]); the common split reduction follows.
#include <stdlib.h>

#include "conv_wgrad.h"

namespace gc {
namespace {

constexpr int kTH = 4, kTW = 32, kUnit = kTH * kTW;      // output rectangle = 128 consecutive values of the reduction index
constexpr int kGyBytes = kUnit * 128;
// S = 2: the 4 x 4 / stride 2 / padding 1 layer.  S = 1: a 3 x 3 / stride 1 / padding 1 layer on an image (BigGAN's first
// discriminator block, BigGAN/model.py:181-200), with the same reads: a filter row is still 4 taps x 4 channels = one block
// of 16 columns per output pixel (the pixels k .. k + 3 of the patch row), the fourth tap's columns and a fourth filter row
// are simply not kept; the patch carries two more columns than the three real taps read.
// KS: filter size -- 4 (S = 2), or with S = 1: 3 (padding 1) or 1 (padding 0: a from-RGB layer, StyleGAN/model.py:453-514; one
// filter row, of whose four columns of taps only the first is kept).
constexpr int patch_h(int s, int ks) { return s == 2 ? 2 * kTH + 2 : kTH + ks - 1; }
constexpr int patch_w(int s) { return s == 2 ? 2 * kTW + 2 : kTW + 4; }
constexpr int patch_bytes(int icp, int s, int ks) { return patch_h(s, ks) * patch_w(s) * 2 * icp; }

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int gy_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

__device__ __forceinline__ bf16x8 read_tr(const char* a0, int second_offset) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + second_offset));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// TX: storage type of the image (uint16_t = bf16, float: rounded to bf16 on the way into the patch)
// ICP: channels of a patch pixel (4, or 8 for 5 to 8 input channels -- the PatchGAN's image pair, Pix2Pix/models.py:225-232: the
// 4 taps x 8 channels of a filter row are then two blocks of 16 columns, twice the MFMAs and accumulators)
template <typename TX, int ICP, int S = 2, int KS = 4>
__global__ __launch_bounds__(kThreads) void conv_wgrad_image_kernel(const WgradParams p) {
  static_assert((S == 2 && KS == 4) || (S == 1 && ICP == 4 && (KS == 3 || KS == 1)), "the stride-1 forms: 4-channel patch pixels, 3 x 3 or 1 x 1");
  constexpr int kPH = patch_h(S, KS), kPW = patch_w(S), NR = KS, kPad = KS == 1 ? 0 : 1;      // NR: filter rows
  constexpr int kPatchBytes = patch_bytes(ICP, S, KS);
  constexpr int NH = ICP / 4;                          // 16-column blocks per filter row
  __shared__ __attribute__((aligned(16))) char smem[2 * (kPatchBytes + kGyBytes)];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int split = blockIdx.x;
  const WgradRange rg = wgrad_range(p, split);
  const TX* x = reinterpret_cast<const TX*>(rg.x);
  const uint16_t* gy = reinterpret_cast<const uint16_t*>(rg.gy);
  const int u0 = rg.kb0 / kUnit, u1 = rg.kend > rg.kb0 ? (rg.kend + kUnit - 1) / kUnit : u0;
  const int tiles_w = p.Wo / kTW;
  const int units_per_sample = (p.Ho / kTH) * tiles_w;
  const int ychunks = p.Cout / 8;                    // 16-byte chunks of a gy pixel

  // ---- what this thread loads per rectangle: 4 chunks of gy, up to 3 patch pixels.  The image values wait in registers as
  // loaded and are rounded to bf16 on their way into LDS, after the multiplications of the rectangle before: a conversion
  // next to its load would put a wait for that load in front of the next load, three memory latencies in a row per rectangle.
  u32x4 yreg[4];
  TX xraw[3][ICP];
  int ppr[3], ppc[3];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const int e = t + i * kThreads;
    ppr[i] = e / kPW;
    ppc[i] = e - ppr[i] * kPW;
  }
  auto load_unit = [&](int u) {
    const int n = u / units_per_sample, rem = u - n * units_per_sample, th = rem / tiles_w, tw = rem - th * tiles_w;
    const size_t k0 = ((size_t)n * p.Ho + (size_t)th * kTH) * p.Wo + (size_t)tw * kTW;      // the rectangle's first output pixel
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = t + i * kThreads, row = idx >> 3, ch = idx & 7;       // row: pixel of the rectangle, 32 per output row
      yreg[i] = u32x4{0u, 0u, 0u, 0u};
      if (ch < ychunks)
        yreg[i] = *reinterpret_cast<const u32x4*>(gy + (k0 + (size_t)(row >> 5) * p.Wo + (row & 31)) * p.y_cstride + p.y_coff + ch * 8);
    }
    const TX* img = x + (size_t)n * p.H * p.W * p.Cin;
    const int ih0 = S * kTH * th - kPad, iw0 = S * kTW * tw - kPad;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int ih = ih0 + ppr[i], iw = iw0 + ppc[i];
#pragma unroll
      for (int c = 0; c < ICP; c++) xraw[i][c] = (TX)0;
      if (t + i * kThreads < kPH * kPW && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
        const TX* src = img + (uint32_t)((ih * p.W + iw) * p.Cin);
        xraw[i][0] = src[0];
#pragma unroll
        for (int c = 1; c < ICP; c++)
          if (p.Cin > c) xraw[i][c] = src[c];
      }
    }
  };
  auto store_unit = [&](int buf) {
    char* pb = smem + buf * (kPatchBytes + kGyBytes);
    char* yb = pb + kPatchBytes;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int idx = t + i * kThreads, row = idx >> 3, ch = idx & 7;
      *reinterpret_cast<u32x4*>(yb + row * 128 + (((ch >> 1) ^ gy_swz(row)) << 5) + (ch & 1) * 16) = yreg[i];
    }
    auto bits = [](TX v) -> uint32_t { return sizeof(TX) == 2 ? (uint32_t)v : (uint32_t)float_to_bf16_bits((float)v); };
#pragma unroll
    for (int i = 0; i < 3; i++)
      if (t + i * kThreads < kPH * kPW) {
#pragma unroll
        for (int h = 0; h < NH; h++)
          *reinterpret_cast<u32x2*>(pb + (t + i * kThreads) * (2 * ICP) + 8 * h) =
              u32x2{bits(xraw[i][4 * h]) | (bits(xraw[i][4 * h + 1]) << 16), bits(xraw[i][4 * h + 2]) | (bits(xraw[i][4 * h + 3]) << 16)};
      }
  };

  f32x4 acc[4][4][NH];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int h = 0; h < NH; h++) acc[i][r][h] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing read: lane (g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3) addresses row 8g + q (and + 4), columns
  // 4pp .. 4pp + 3 of a 16-column block and receives, for column lane & 15, the rows 8g .. 8g + 7
  const int k0 = 8 * (lane >> 4) + ((lane >> 2) & 3), pp8 = (lane & 3) * 8;
  const int co_blocks = p.Cout / 16;

  if (u0 < u1) {
    load_unit(u0);
    store_unit(0);
  }
  __syncthreads();
  for (int u = u0; u < u1; u++) {
    const int buf = (u - u0) & 1;
    const bool more = u + 1 < u1;
    if (more) load_unit(u + 1);
    const char* pb = smem + buf * (kPatchBytes + kGyBytes);
    const char* yb = pb + kPatchBytes;
    bf16x8 a[4], b[4][NH];
    const int yrow = wave * kTW + k0;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < co_blocks) a[i] = read_tr(yb + yrow * 128 + ((i ^ gy_swz(k0)) << 5) + pp8, 4 * 128);
    // output pixel k of the row meets the patch pixels 2k .. 2k + 3 of filter row r: 4 taps x ICP channels, contiguous
    // (stride 1: the patch pixels k .. k + 3, the fourth being the unused tap)
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int h = 0; h < NH; h++) b[r][h] = read_tr(pb + ((S * wave + r) * kPW + S * k0) * (2 * ICP) + 32 * h + pp8, 4 * S * 2 * ICP);
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < co_blocks)
#pragma unroll
        for (int r = 0; r < NR; r++)
#pragma unroll
          for (int h = 0; h < NH; h++) acc[i][r][h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[r][h], acc[i][r][h], 0, 0, 0);
    if (more) store_unit(buf ^ 1);
    __syncthreads();
  }

  // ---- the four waves' tiles -> one: waves 2, 3 hand theirs to waves 0, 1, then wave 1 to wave 0 (one column block at a time)
  float* red = reinterpret_cast<float*>(smem);          // two slots of 16 fragments x 64 lanes x 4 floats (16 KB each)
#pragma unroll
  for (int h = 0; h < NH; h++) {
    auto slot_io = [&](int slot, bool write) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          f32x4* q = reinterpret_cast<f32x4*>(red + ((size_t)slot * 16 + i * 4 + r) * 256 + lane * 4);
          if (write) *q = acc[i][r][h];
          else acc[i][r][h] += *q;
        }
    };
    if (h > 0) __syncthreads();                           // the slots of the block before have been read
    if (wave >= 2) slot_io(wave - 2, true);
    __syncthreads();
    if (wave < 2) slot_io(wave, false);
    __syncthreads();
    if (wave == 1) slot_io(0, true);
    __syncthreads();
    if (wave == 0) slot_io(0, false);
  }
  if (wave != 0) return;
  // [split][Cout][tap][ICP]: column (lane & 15) of block h of filter row r is tap 4 r + (16 h + column) / ICP, channel column % ICP
  // (stride 1: [split][Cout][3 x 3 taps][4]; column (lane & 15) is tap column (lane & 15) / 4, the fourth is dropped)
  constexpr int kRow = KS * KS * ICP, kTapsPerRow = KS;
  float* part = p.part + (size_t)split * p.Cout * kRow;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < NR; r++)
#pragma unroll
      for (int h = 0; h < NH; h++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int co = i * 16 + (lane >> 4) * 4 + e;
          if (co < p.Cout && (S == 2 || (lane & 15) < 4 * KS)) part[(size_t)co * kRow + r * (kTapsPerRow * ICP) + h * 16 + (lane & 15)] = acc[i][r][h][e];
        }
}

}  // namespace

// bf16 operands, dense image of at most 8 channels, 4 x 4 / stride 2 / padding 1, output columns in 32s, output rows in
// fours, 16 / 32 / 48 / 64 output channels with 16-byte aligned gy pixels; splits that start on a rectangle.
bool wgrad_image_applies(const WgradParams& p, const WgradLaunch& l) {
  if (!p.image || !l.bf16 || l.x3 || !l.aligned) return false;
  return p.k_per_split % kUnit == 0 && p.y_cstride % 8 == 0 && p.y_coff % 8 == 0;
}

static bool wgrad_image_stride1(const WgradParams& p) {
  static const bool off = getenv("GC_NO_IMAGE_3X3") != nullptr;
  const bool k3 = p.kh == 3 && p.kw == 3 && p.ph == 1 && p.pw == 1, k1 = p.kh == 1 && p.kw == 1 && p.ph == 0 && p.pw == 0;
  return !off && p.thin && p.Cin <= 4 && (k3 || k1) && p.sh == 1 && p.sw == 1 && p.Wo % kTW == 0 &&
         p.Ho % kTH == 0 && p.H == p.Ho && p.W == p.Wo && p.Cout % 16 == 0 && p.Cout <= 64 && p.x_cstride == p.Cin && p.x_coff == 0 &&
         (long)p.H * p.W * p.Cin < (1l << 31);
}

bool wgrad_image_geometry(const WgradParams& p) {
  if (wgrad_image_stride1(p)) return true;
  return p.thin && p.Cin <= 8 && p.kh == 4 && p.kw == 4 && p.sh == 2 && p.sw == 2 && p.ph == 1 && p.pw == 1 && p.Wo % kTW == 0 &&
         p.Ho % kTH == 0 && p.H == 2 * p.Ho && p.W == 2 * p.Wo && p.Cout % 16 == 0 && p.Cout <= 64 && p.x_cstride == p.Cin &&
         p.x_coff == 0 && (long)p.H * p.W * p.Cin < (1l << 31);
}

int launch_wgrad_image(const WgradParams& p, const WgradLaunch& l, hipStream_t stream) {
  if (p.K <= 0 || p.splits <= 0) return GC_OK;
  ProfScope ps("gc::conv_wgrad_image_kernel", l.flops, stream);
  const dim3 grid((unsigned)p.splits), block(kThreads);
  if (wgrad_image_stride1(p)) {
    if (p.kh == 3) {
      if (p.x_f32) hipLaunchKernelGGL((conv_wgrad_image_kernel<float, 4, 1, 3>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_image_kernel<uint16_t, 4, 1, 3>), grid, block, 0, stream, p);
    } else {
      if (p.x_f32) hipLaunchKernelGGL((conv_wgrad_image_kernel<float, 4, 1, 1>), grid, block, 0, stream, p);
      else hipLaunchKernelGGL((conv_wgrad_image_kernel<uint16_t, 4, 1, 1>), grid, block, 0, stream, p);
    }
  } else if (p.ICp == 8) {
    if (p.x_f32) hipLaunchKernelGGL((conv_wgrad_image_kernel<float, 8>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_image_kernel<uint16_t, 8>), grid, block, 0, stream, p);
  } else {
    if (p.x_f32) hipLaunchKernelGGL((conv_wgrad_image_kernel<float, 4>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_wgrad_image_kernel<uint16_t, 4>), grid, block, 0, stream, p);
  }
  return check_hip(hipGetLastError(), "conv_wgrad_image launch");
}

}  // namespace gc
