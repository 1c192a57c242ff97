// Convolution passes with very few output channels (RGB images out of a generator's last transposed conv, the data
// gradient of a discriminator's first conv, 3x3 to-image layers): these are bound by reading the input, and a
// gather-GEMM tile would fetch every input pixel once per filter tap and pixel class.  Here a workgroup owns a
// rectangle of output pixels of ONE image, stages the input patch that feeds it (halo included) once, and works
// in two phases:
//   1. P[(tap, oc)][q] = sum_c w[oc][tap][c] * in[q][c]   for every patch pixel q  -- a small MFMA GEMM with the
//      (tap, output channel) pairs on the rows (at most 64) and the patch pixels on the columns;
//   2. out[pixel][oc] = sum over the taps of the pixel's class of P[(tap, oc)][q(pixel, tap)]  (+ bias, activation)
//      read back from LDS, written with consecutive lanes on consecutive output elements.
// Padding is a zero row of the patch, so phase 2 needs no bounds logic on the input side.  bf16 MFMA precision only.
#include <stdio.h>

#include "conv_gather.h"

namespace gc {

namespace {

constexpr int kTW = 16;               // class-local output rectangle of a workgroup: TH (8, or 4 where the halo of a 4 x 4
                                     // filter makes the patch of 8 rows too large) x 16
constexpr int kMaxPatch = 192;       // patch pixels (rows of the staged input tile)
constexpr int kMaxRows = 96;         // (tap, oc) pairs (64 in the MR = 4 build, 96 in the MR = 6 one: 16 taps x 6 channels)
constexpr int kPStride = 196;        // floats per row of P: 4 * 196 = 16 (mod 64) keeps the fragment stores conflict-free

__device__ __forceinline__ int sswz(int row) { return (row >> 1) & 7; }

}  // namespace

// MR: 16-row fragments of (tap, channel) pairs the build holds (4: up to 64 pairs, 6: up to 96)
template <typename TIn, typename TOut, int TH, int MR = 4>
__global__ __launch_bounds__(kThreads) void conv_scatter_kernel(const GatherParams p) {
  constexpr int kTH = TH;
  constexpr int EPC = 16 / (int)sizeof(TIn);          // elements per 16-byte source chunk
  constexpr int CPR = kBK / EPC;                      // source chunks per 64-channel row
  constexpr int RPP = kThreads / CPR;                 // rows loaded per pass
  constexpr int PASSES = kMaxPatch / RPP;
  constexpr int X_BYTES = kMaxPatch * 128;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xin = smem;                       // [patch pixel][64 ch] bf16, 16-byte chunks XOR-swizzled by (row >> 1) & 7
  char* wts = smem + X_BYTES;             // [(tap, oc)][64 ch] bf16, same swizzle
  float* P = reinterpret_cast<float*>(smem);                       // phase 2 (aliases the tiles)
  __shared__ int tap_e[kMaxTaps];
  __shared__ int cls_start[kMaxClasses + 1];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const uint32_t tiles = (uint32_t)p.sc_tiles_h * (uint32_t)p.sc_tiles_w;
  const uint32_t lid = xcd_remap(blockIdx.x, (uint32_t)p.N * tiles);
  const int n = lid / tiles;
  const int th = (lid % tiles) / p.sc_tiles_w, tw = (lid % tiles) % p.sc_tiles_w;
  const int PW = p.sc_pw, NP = p.sc_ph * p.sc_pw;
  const int ih0 = th * kTH * p.a + p.sc_dh_min, iw0 = tw * kTW * p.a + p.sc_dw_min;
  const int R = p.T * p.OC, RF = (R + 15) >> 4, NF = (NP + 15) >> 4;

  if (t < kMaxTaps) tap_e[t] = ((int)p.tt.dh[t] & 0xff) | (((int)p.tt.dw[t] & 0xff) << 8) | ((int)p.tt.wtap[t] << 16);
  if (t <= kMaxClasses) cls_start[t] = p.tt.cls_tap_start[t];

  f32x4 acc[MR][3];
#pragma unroll
  for (int i = 0; i < MR; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const TIn* in = reinterpret_cast<const TIn*>(p.in);
  const int chunk = t % CPR, row0 = t / CPR;
  for (int c0 = 0; c0 < p.ICp; c0 += kBK) {
    // ---- stage the patch (converted to bf16) and the weight rows of this channel step
    u32x4 reg[PASSES];
#pragma unroll
    for (int i = 0; i < PASSES; i++) {
      const int q = row0 + RPP * i;
      const int pr = q / PW, pc = q - pr * PW;
      const int ih = ih0 + pr, iw = iw0 + pc;
      const int c = c0 + chunk * EPC;
      const bool ok = q < NP && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW && c < p.IC;
      reg[i] = u32x4{0u, 0u, 0u, 0u};
      if (ok) reg[i] = *reinterpret_cast<const u32x4*>(in + (((size_t)n * p.IH + ih) * p.IW + iw) * p.in_cstride + p.in_coff + c);
    }
    u32x4 wreg[MR / 2];
#pragma unroll
    for (int i = 0; i < MR / 2; i++) {
      const int idx = t + kThreads * i;
      const int row = idx >> 3, ch = idx & 7;
      wreg[i] = u32x4{0u, 0u, 0u, 0u};
      if (row < R) {
        const int tap = row / p.OC, oc = row - tap * p.OC;
        wreg[i] = *reinterpret_cast<const u32x4*>(p.wp + (size_t)oc * p.row_pitch + tap * p.ICp + c0 + ch * 8);
      }
    }
    __syncthreads();   // the previous step's fragments have been read
#pragma unroll
    for (int i = 0; i < PASSES; i++) {
      const int q = row0 + RPP * i;
      if (sizeof(TIn) == 4) {
        const u32x2 v = cvt_bf16x4(__builtin_bit_cast(f32x4, reg[i]));
        *reinterpret_cast<u32x2*>(xin + q * 128 + (((chunk >> 1) ^ sswz(q)) << 4) + (chunk & 1) * 8) = v;
      } else {
        *reinterpret_cast<u32x4*>(xin + q * 128 + ((chunk ^ sswz(q)) << 4)) = reg[i];
      }
    }
#pragma unroll
    for (int i = 0; i < MR / 2; i++) {
      const int idx = t + kThreads * i;
      const int row = idx >> 3, ch = idx & 7;
      *reinterpret_cast<u32x4*>(wts + row * 128 + ((ch ^ sswz(row)) << 4)) = wreg[i];
    }
    __syncthreads();
    // ---- P += W * patch^T : wave w owns the column fragments w, w+4, w+8
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int kc = ks * 4 + (lane >> 4);
      bf16x8 a[MR], b[3];
#pragma unroll
      for (int i = 0; i < MR; i++) {
        const int row = i * 16 + (lane & 15);
        a[i] = *reinterpret_cast<const bf16x8*>(wts + row * 128 + ((kc ^ sswz(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int row = (wave + 4 * j) * 16 + (lane & 15);
        b[j] = *reinterpret_cast<const bf16x8*>(xin + row * 128 + ((kc ^ sswz(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MR; i++)
        if (i < RF)
#pragma unroll
          for (int j = 0; j < 3; j++)
            if (wave + 4 * j < NF) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // all fragment reads done: P may overwrite the tiles
#pragma unroll
  for (int i = 0; i < MR; i++)
    if (i < RF)
#pragma unroll
      for (int j = 0; j < 3; j++)
        if (wave + 4 * j < NF)
#pragma unroll
          for (int e = 0; e < 4; e++)
            P[(i * 16 + (lane >> 4) * 4 + e) * kPStride + (wave + 4 * j) * 16 + (lane & 15)] = acc[i][j][e];
  __syncthreads();

  // ---- phase 2: consecutive threads take consecutive (pixel, channel) elements of each output row
  const int b = p.b, EW = kTW * b, EH = kTH * b;
  const int ew_shift = b == 1 ? 4 : 5, b_shift = b == 1 ? 0 : 1;
  const int total = EH * EW * p.OC;
  TOut* out = reinterpret_cast<TOut*>(p.out);
  for (int e = t; e < total; e += kThreads) {
    const uint32_t pixl = fd_div((uint32_t)e, p.sc_div_oc);
    const int oc = e - (int)pixl * p.OC;
    const int ow_l = (int)pixl & (EW - 1), oh_l = (int)pixl >> ew_shift;
    const int oh = th * EH + oh_l, ow = tw * EW + ow_l;
    if (oh >= p.OH || ow >= p.OW) continue;
    const int cls = (oh_l & (b - 1)) * b + (ow_l & (b - 1));
    const int ohc = oh_l >> b_shift, owc = ow_l >> b_shift;     // class-local coordinates inside the rectangle
    float sum = 0.f;
    for (int tp = cls_start[cls]; tp < cls_start[cls + 1]; tp++) {
      const int en = tap_e[tp];
      const int dh = (int)(int8_t)(en & 0xff), dw = (int)(int8_t)((en >> 8) & 0xff), wt = en >> 16;
      const int q = (ohc * p.a + dh - p.sc_dh_min) * PW + (owc * p.a + dw - p.sc_dw_min);
      sum += P[(wt * p.OC + oc) * kPStride + q];
    }
    if (p.bias != nullptr) sum += p.bias[oc];
    sum = apply_act(sum, p.act, p.slope);
    const size_t o = (((size_t)n * p.OH + oh) * p.OW + ow) * p.out_cstride + p.out_coff + oc;
    if (sizeof(TOut) == 4) reinterpret_cast<float*>(out)[o] = sum;
    else reinterpret_cast<uint16_t*>(out)[o] = float_to_bf16_bits(sum);
  }
}

// Fills the patch geometry when the problem suits this kernel (stride-1 gather, up to 64 (tap, channel) rows,
// a halo of at most two pixels, vector-aligned channels); returns false otherwise.
bool plan_scatter(GatherParams& p, const GatherLaunch& l) {
  if (l.x3 || l.mode != kVecLoads || p.ch_sum != nullptr || p.a != 1 || (p.b != 1 && p.b != 2)) return false;
  if (p.T * p.OC > kMaxRows || p.OC > 16) return false;
  if ((long)p.OH * p.OW < 256) return false;   // tiny images: the tile would be mostly halo
  const int n = p.tt.cls_tap_start[p.tt.n_classes];
  if (n <= 0 || p.tt.n_classes != p.b * p.b) return false;
  int dh0 = 127, dh1 = -128, dw0 = 127, dw1 = -128;
  for (int e = 0; e < n; e++) {
    if (p.tt.dh[e] < dh0) dh0 = p.tt.dh[e];
    if (p.tt.dh[e] > dh1) dh1 = p.tt.dh[e];
    if (p.tt.dw[e] < dw0) dw0 = p.tt.dw[e];
    if (p.tt.dw[e] > dw1) dw1 = p.tt.dw[e];
  }
  for (int c = 0; c < p.tt.n_classes; c++)   // phase 2 derives the class from the pixel's parity
    if (p.tt.cls_ph[c] != c / p.b || p.tt.cls_pw[c] != c % p.b) return false;
  const int pw = (kTW - 1) * p.a + (dw1 - dw0) + 1;
  int th = 8, ph = (th - 1) * p.a + (dh1 - dh0) + 1;
  if (ph * pw > kMaxPatch) { th = 4; ph = (th - 1) * p.a + (dh1 - dh0) + 1; }
  if (ph * pw > kMaxPatch) return false;
  p.sc_dh_min = dh0; p.sc_dw_min = dw0; p.sc_ph = ph; p.sc_pw = pw; p.sc_th = th;
  p.sc_tiles_h = ((p.OH + p.b - 1) / p.b + th - 1) / th;
  p.sc_tiles_w = ((p.OW + p.b - 1) / p.b + kTW - 1) / kTW;
  p.sc_div_oc = make_fastdiv((uint32_t)p.OC);
  return true;
}

template <typename TIn, typename TOut>
static int launch_scatter_one(const GatherParams& p, double flops, hipStream_t stream) {
  const int rows16 = (p.T * p.OC + 15) / 16 * 16;
  int smem = rows16 * kPStride * 4;
  if (smem < (kMaxPatch + kMaxRows) * 128) smem = (kMaxPatch + kMaxRows) * 128;
  const bool wide = rows16 > 64;
  auto kern = wide ? (p.sc_th == 8 ? conv_scatter_kernel<TIn, TOut, 8, 6> : conv_scatter_kernel<TIn, TOut, 4, 6>)
                   : (p.sc_th == 8 ? conv_scatter_kernel<TIn, TOut, 8, 4> : conv_scatter_kernel<TIn, TOut, 4, 4>);
  if (smem > 64 * 1024 && ensure_dynamic_smem(reinterpret_cast<const void*>(kern), smem) != GC_OK) return GC_HIP_ERROR;
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_scatter_kernel<%s, %s>", sizeof(TIn) == 4 ? "float" : "unsigned short",
           sizeof(TOut) == 4 ? "float" : "unsigned short");
  ProfScope ps(name, flops, stream);
  const unsigned nblocks = (unsigned)(p.N * p.sc_tiles_h * p.sc_tiles_w);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kThreads), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_scatter launch");
}

int launch_scatter(const GatherParams& p, const GatherLaunch& l, bool out_f32, hipStream_t s) {
  if (p.N == 0) return GC_OK;
  if (l.in_bf16) return out_f32 ? launch_scatter_one<uint16_t, float>(p, l.flops, s) : launch_scatter_one<uint16_t, uint16_t>(p, l.flops, s);
  return out_f32 ? launch_scatter_one<float, float>(p, l.flops, s) : launch_scatter_one<float, uint16_t>(p, l.flops, s);
}

}  // namespace gc
