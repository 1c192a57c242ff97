// Implicit-GEMM convolution on the bf16 matrix cores: forward pass and data-gradient pass in one kernel.
//
//   out[pix][oc] = sum over (tap, ic) of  in[gather(pix, tap)][ic] * wp[oc][tap][ic]
//
// GEMM view: rows = output channels (MFMA A operand, from the packed weights), columns = output pixels (MFMA B
// operand, gathered from the NHWC activation on the fly: no im2col buffer), reduction = (tap, input channel).
// With channels on the MFMA row axis each lane ends up holding 4 consecutive output channels of one pixel, i.e.
// one 16-byte NHWC store.
//
// Block = 256 threads = 4 waves, tile TO channels x TP pixels, reduction depth 64 per step, operands staged
// global -> registers -> LDS (bf16, XOR-swizzled 128-byte rows, conflict-free ds_read_b128) with two LDS
// buffers: the loads of step k+1 are in flight while step k runs on the matrix cores.
//
// Load modes: 16-byte vector loads (channel counts / strides that are multiples of a chunk), element loads, and a
// thin-input mode for up to 32 input channels (RGB images, 6-channel image pairs, scores) in which the reduction
// runs over (tap, channel) jointly -- 64 per step -- instead of padding every tap to 64 channels.
// Small problems (few tiles, long reduction: 4x4 -> 1x1 "full window" layers) split the reduction over several
// workgroups that add their partial tiles into a zeroed fp32 output.
#include <stdio.h>

#include <mutex>
#include <stdlib.h>

#include "conv_gather.h"

namespace gc {

template <int TO, int TP>
struct TileCfg {
  static constexpr int WO = (TO >= 64) ? 2 : 1;      // waves along channels
  static constexpr int WP = 4 / WO;                  // waves along pixels
  static constexpr int FO = TO / WO / 16;            // 16x16 fragments per wave, channel axis
  static constexpr int FP = TP / WP / 16;            // pixel axis
};

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <typename TIn, int TO, int TP, bool X3, int MODE>
__global__ __launch_bounds__(kThreads) void conv_gather_kernel(const GatherParams p) {
  constexpr bool ALIGNED = MODE == kVecLoads;
  constexpr bool THIN = MODE == kThinInput;
  using C = TileCfg<TO, TP>;
  constexpr int PL = X3 ? 2 : 1;
  constexpr int EPC = 16 / (int)sizeof(TIn);   // elements per 16-byte chunk of the activation
  constexpr int CPR = kBK / EPC;               // chunks per activation row per step
  constexpr int AR = TP / 64;                  // activation rows per thread
  constexpr int AC = CPR / 4;                  // activation chunks per thread per row
  constexpr int WR = (TO >= 32) ? TO / 32 : 1; // weight rows per thread
  constexpr int W_BYTES = TO * 128, A_BYTES = TP * 128;
  constexpr int STAGE = PL * (W_BYTES + A_BYTES);

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tap_lds = reinterpret_cast<int*>(smem + 2 * STAGE);   // thin-input mode: (dh, dw) per tap

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wo = wave / C::WP, wp_ = wave % C::WP;

  // ---- which tile ------------------------------------------------------------------------------------
  const uint32_t nblocks = (uint32_t)p.n_pix_tiles * (uint32_t)p.n_oc_tiles * (uint32_t)p.ksplit;
  uint32_t lid = xcd_remap(blockIdx.x, nblocks);
  const int split = lid % p.ksplit;
  lid /= p.ksplit;
  const int pix_tile = lid / p.n_oc_tiles;
  const int oc0 = (lid % p.n_oc_tiles) * TO;
  int cls = 0, m0;
  if (p.cls_interleave) {   // equal-sized classes: (tile, class) order, the classes of one image region are neighbours
    cls = pix_tile % p.tt.n_classes;
    m0 = (pix_tile / p.tt.n_classes) * TP;
  } else {
#pragma unroll
    for (int c = 1; c < kMaxClasses; c++)
      if (c < p.tt.n_classes && pix_tile >= p.tt.cls_tile_start[c]) cls = c;
    m0 = (pix_tile - p.tt.cls_tile_start[cls]) * TP;
  }
  const int OHc = p.tt.cls_oh[cls], OWc = p.tt.cls_ow[cls];
  const int Mc = p.N * OHc * OWc;
  const int tap_begin = p.tt.cls_tap_start[cls], tap_end = p.tt.cls_tap_start[cls + 1];

  // ---- per-thread gather rows --------------------------------------------------------------------------
  int row_base[AR], row_ih[AR], row_iw[AR];
  bool row_ok[AR];
#pragma unroll
  for (int i = 0; i < AR; i++) {
    const int m = m0 + (t >> 2) + 64 * i;
    row_ok[i] = m < Mc;
    const int mm = row_ok[i] ? m : 0;
    const int ow = mm % OWc, tmp = mm / OWc;
    const int oh = tmp % OHc, n = tmp / OHc;
    row_base[i] = n * p.IH * p.IW;
    row_ih[i] = oh * p.a;
    row_iw[i] = ow * p.a;
  }
  const TIn* in = reinterpret_cast<const TIn*>(p.in);
  if (THIN) {
    if (t < kMaxTaps) tap_lds[t] = ((int)p.tt.dh[t] & 0xffff) | ((int)p.tt.dw[t] << 16);
    __syncthreads();
  }

  struct Regs {
    u32x4 a[AR][AC];
    u32x4 w[PL][WR];
  };

  // thin-input mode: `tap` is the reduction step; chunk ch of a row holds elements step*64 + ch*EPC .. of the
  // joint (tap, channel) index, each aligned group of 4 belonging to one tap
  auto load_tile_thin = [&](Regs& R, int step) {
#pragma unroll
    for (int i = 0; i < AR; i++) {
#pragma unroll
      for (int j = 0; j < AC; j++) {
        u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int sub = 0; sub < EPC / 4; sub++) {
          const int kk = step * kBK + ((t & 3) + 4 * j) * EPC + sub * 4;
          const int tapi = kk >> p.icp_log2, c = kk & (p.ICp - 1);
          if (tapi < p.T && row_ok[i]) {
            const int e = tap_lds[tapi];
            const int ih = row_ih[i] + (int)(int16_t)(e & 0xffff), iw = row_iw[i] + (e >> 16);
            if ((unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW) {
              const TIn* src = in + ((size_t)(row_base[i] + ih * p.IW + iw) * p.in_cstride + p.in_coff + c);
#pragma unroll
              for (int q = 0; q < 4; q++)
                if (c + q < p.IC) {
                  if (sizeof(TIn) == 4) v[q] = reinterpret_cast<const uint32_t*>(src)[q];
                  else v[sub * 2 + (q >> 1)] |= (uint32_t)reinterpret_cast<const uint16_t*>(src)[q] << (16 * (q & 1));
                }
            }
          }
        }
        R.a[i][j] = v;
      }
    }
    if (TO >= 32 || t < TO * 8) {
#pragma unroll
      for (int pl = 0; pl < PL; pl++)
#pragma unroll
        for (int i = 0; i < WR; i++) {
          const int oc = oc0 + (t >> 3) + 32 * i;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (oc < p.OC) v = *reinterpret_cast<const u32x4*>(p.wp + pl * p.plane_stride + (size_t)oc * p.row_pitch + step * kBK + (t & 7) * 8);
          R.w[pl][i] = v;
        }
    }
  };

  auto load_tile = [&](Regs& R, int tap, int c0) {
    if (THIN) { load_tile_thin(R, tap); return; }
    const int dh = p.tt.dh[tap], dw = p.tt.dw[tap], wt = p.tt.wtap[tap];
#pragma unroll
    for (int i = 0; i < AR; i++) {
      const int ih = row_ih[i] + dh, iw = row_iw[i] + dw;
      const bool ok = row_ok[i] && (unsigned)ih < (unsigned)p.IH && (unsigned)iw < (unsigned)p.IW;
      const TIn* src = in + ((size_t)(row_base[i] + ih * p.IW + iw) * p.in_cstride + p.in_coff + c0);
#pragma unroll
      for (int j = 0; j < AC; j++) {
        const int c = ((t & 3) + 4 * j) * EPC;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ALIGNED) {
          if (ok && c0 + c < p.IC) v = *reinterpret_cast<const u32x4*>(src + c);
        } else {
          if (ok) {
            if (sizeof(TIn) == 4) {
#pragma unroll
              for (int e = 0; e < 4; e++)
                if (c0 + c + e < p.IC) v[e] = reinterpret_cast<const uint32_t*>(src)[c + e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; e++)
                if (c0 + c + e < p.IC) v[e >> 1] |= (uint32_t)reinterpret_cast<const uint16_t*>(src)[c + e] << (16 * (e & 1));
            }
          }
        }
        R.a[i][j] = v;
      }
    }
    if (TO >= 32 || t < TO * 8) {
#pragma unroll
      for (int pl = 0; pl < PL; pl++)
#pragma unroll
        for (int i = 0; i < WR; i++) {
          const int oc = oc0 + (t >> 3) + 32 * i;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (oc < p.OC)
            v = *reinterpret_cast<const u32x4*>(p.wp + pl * p.plane_stride + (size_t)oc * p.row_pitch + wt * p.ICp + c0 + (t & 7) * 8);
          R.w[pl][i] = v;
        }
    }
  };

  auto store_tile = [&](const Regs& R, int buf) {
    char* wb = smem + buf * STAGE;
    char* ab = wb + PL * W_BYTES;
#pragma unroll
    for (int i = 0; i < AR; i++) {
      const int row = (t >> 2) + 64 * i;
#pragma unroll
      for (int j = 0; j < AC; j++) {
        const int ch = (t & 3) + 4 * j;
        if (sizeof(TIn) == 4) {
          const f32x4 v = __builtin_bit_cast(f32x4, R.a[i][j]);
          const u32x2 hi = cvt_bf16x4(v);
          const int off = row * 128 + (((ch >> 1) ^ swz(row)) << 4) + (ch & 1) * 8;
          *reinterpret_cast<u32x2*>(ab + off) = hi;
          if (X3) *reinterpret_cast<u32x2*>(ab + A_BYTES + off) = cvt_bf16x4(v - widen_bf16x4(hi));
        } else {
          *reinterpret_cast<u32x4*>(ab + row * 128 + ((ch ^ swz(row)) << 4)) = R.a[i][j];
        }
      }
    }
    if (TO >= 32 || t < TO * 8) {
#pragma unroll
      for (int pl = 0; pl < PL; pl++)
#pragma unroll
        for (int i = 0; i < WR; i++) {
          const int row = (t >> 3) + 32 * i;
          *reinterpret_cast<u32x4*>(wb + pl * W_BYTES + row * 128 + (((t & 7) ^ swz(row)) << 4)) = R.w[pl][i];
        }
    }
  };

  f32x4 acc[C::FO][C::FP];
#pragma unroll
  for (int i = 0; i < C::FO; i++)
#pragma unroll
    for (int j = 0; j < C::FP; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const char* wb = smem + buf * STAGE;
    const char* ab = wb + PL * W_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int kc = ks * 4 + (lane >> 4);
      bf16x8 a[PL][C::FO], b[PL][C::FP];
#pragma unroll
      for (int f = 0; f < C::FO; f++) {
        const int row = wo * (TO / C::WO) + f * 16 + (lane & 15);
        const int off = row * 128 + ((kc ^ swz(row)) << 4);
#pragma unroll
        for (int pl = 0; pl < PL; pl++) a[pl][f] = *reinterpret_cast<const bf16x8*>(wb + pl * W_BYTES + off);
      }
#pragma unroll
      for (int f = 0; f < C::FP; f++) {
        const int row = wp_ * (TP / C::WP) + f * 16 + (lane & 15);
        const int off = row * 128 + ((kc ^ swz(row)) << 4);
#pragma unroll
        for (int pl = 0; pl < PL; pl++) b[pl][f] = *reinterpret_cast<const bf16x8*>(ab + pl * A_BYTES + off);
      }
#pragma unroll
      for (int i = 0; i < C::FO; i++)
#pragma unroll
        for (int j = 0; j < C::FP; j++) {
          if (X3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
        }
    }
  };

  // ---- main loop ---------------------------------------------------------------------------------------
  const int csteps = THIN ? 1 : p.ICp / kBK;
  const int total_steps = THIN ? (p.T * p.ICp + kBK - 1) / kBK : (tap_end - tap_begin) * csteps;
  const int per_split = (total_steps + p.ksplit - 1) / p.ksplit;
  const int step0 = split * per_split;
  const int nsteps = max(0, min(total_steps, step0 + per_split) - step0);
  int tap = THIN ? step0 : tap_begin + step0 / csteps, cs = THIN ? 0 : step0 % csteps;
  auto advance = [&]() {
    if (++cs == csteps) { cs = 0; tap++; }
  };
  {
    Regs R;
    if (nsteps > 0) {
      load_tile(R, tap, cs * kBK);
      store_tile(R, 0);
    }
    __syncthreads();
    for (int it = 0; it < nsteps; it++) {
      const bool more = it + 1 < nsteps;
      if (more) { advance(); load_tile(R, tap, cs * kBK); }
      compute(it & 1);
      if (more) store_tile(R, (it + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  const bool vec_ok = p.out_vec != 0;
  float ssum[C::FO][4], ssq[C::FO][4];
#pragma unroll
  for (int i = 0; i < C::FO; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) ssum[i][r] = ssq[i][r] = 0.f;
  const bool stats = p.ch_sum != nullptr || p.stat_rows != nullptr;
  const bool of32 = p.out_f32 != 0;
  const int osz = of32 ? 4 : 2;
#pragma unroll
  for (int j = 0; j < C::FP; j++) {
    const int m = m0 + wp_ * (TP / C::WP) + j * 16 + (lane & 15);
    const bool mok = m < Mc;
    const int mm = mok ? m : 0;
    const int ow = mm % OWc, tmp = mm / OWc;
    const int oh = tmp % OHc, n = tmp / OHc;
    const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
    const size_t eoff = pix * p.out_cstride + p.out_coff;      // element offset of this pixel's channel slice
    char* dst = reinterpret_cast<char*>(p.out) + eoff * osz;
#pragma unroll
    for (int i = 0; i < C::FO; i++) {
      const int oc = oc0 + wo * (TO / C::WO) + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float x = acc[i][j][r];
        if (p.pre_scale != nullptr && oc + r < p.OC) x *= p.pre_scale[(size_t)n * p.OC + oc + r];
        if (p.bias != nullptr && oc + r < p.OC && split == 0) x += p.bias[oc + r];
        x = apply_act(x, p.act, p.slope);
        if (p.post_scale != nullptr && oc + r < p.OC) x *= p.post_scale[(size_t)n * p.OC + oc + r];
        if (!of32) x = bf16_bits_to_float(float_to_bf16_bits(x));
        v[r] = x;
        if (stats && mok && oc + r < p.OC) { ssum[i][r] += x; ssq[i][r] += x * x; }
      }
      if (!mok) continue;
      if (p.ksplit > 1) {   // partial tile of a split reduction: fp32 adds into the zeroed output
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oc + r < p.OC) atomicAdd(reinterpret_cast<float*>(dst) + oc + r, v[r]);
        continue;
      }
      if (vec_ok && oc + 3 < p.OC) {
        if (of32) {
          *reinterpret_cast<f32x4*>(dst + (size_t)oc * 4) = f32x4{v[0], v[1], v[2], v[3]};
          if (p.out_bf16 != nullptr) *reinterpret_cast<u32x2*>(p.out_bf16 + eoff + oc) = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]});
        } else {
          *reinterpret_cast<u32x2*>(dst + (size_t)oc * 2) = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]});
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oc + r < p.OC) {
            if (of32) reinterpret_cast<float*>(dst)[oc + r] = v[r];
            else reinterpret_cast<uint16_t*>(dst)[oc + r] = float_to_bf16_bits(v[r]);
          }
      }
    }
  }
  if (stats) {
    // sum over the 16 pixel lanes that share a channel quad, then one atomic per channel per wave
#pragma unroll
    for (int i = 0; i < C::FO; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float s = ssum[i][r], q = ssq[i][r];
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
          s += __shfl_xor(s, d, 64);
          q += __shfl_xor(q, d, 64);
        }
        const int oc = oc0 + wo * (TO / C::WO) + i * 16 + (lane >> 4) * 4 + r;
        if ((lane & 15) == 0 && oc < p.OC) {
          if (p.stat_rows != nullptr) {      // this wave's pixels of this tile are one row of the partial table
            const size_t row = (size_t)((p.tt.cls_tile_start[cls] + m0 / TP) * C::WP + wp_);
            *reinterpret_cast<float2*>(p.stat_rows + (row * (size_t)p.OC + oc) * 2) = float2{s, q};
          } else {
            atomicAdd(p.ch_sum + oc, s);
            if (p.ch_sumsq != nullptr) atomicAdd(p.ch_sumsq + oc, q);
          }
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
PackGeom pack_geom(int ic, int taps, bool allow_thin) {
  PackGeom g;
  if (allow_thin && ic <= 32) {
    g.icp = 4;
    while (g.icp < ic) g.icp *= 2;
    g.row_pitch = (taps * g.icp + kBK - 1) / kBK * kBK;
    g.thin = 1;
  } else {
    g.icp = (ic + kBK - 1) / kBK * kBK;
    g.row_pitch = taps * g.icp;
    g.thin = 0;
  }
  return g;
}

void plan_gather(GatherParams& p, GatherLaunch& l) {
  // Channel tile: the smallest of {16, 64, 128} covering OC (128 when OC > 64); pixel tile 128, or 256 for the
  // 16-channel tile (thin outputs such as RGB images or critic scores).
  l.to = p.OC <= 16 ? 16 : (p.OC <= 64 ? 64 : 128);
  l.tp = l.to == 16 ? 256 : 128;
  if (l.to == 128) {   // too few 128-channel tiles to cover the CUs: halve the channel tile
    long pix_tiles = 0;
    for (int c = 0; c < p.tt.n_classes; c++) pix_tiles += ((long)p.N * p.tt.cls_oh[c] * p.tt.cls_ow[c] + l.tp - 1) / l.tp;
    static const long to64_max = getenv("GC_TO64_MAX_TILES") ? atol(getenv("GC_TO64_MAX_TILES")) : 128;
    if (pix_tiles * ((p.OC + 127) / 128) <= to64_max) l.to = 64;
  }
  p.n_oc_tiles = (p.OC + l.to - 1) / l.to;
  // LDS-DMA path, large problems: 128 channels x 256 pixels per 8-wave workgroup (one per CU, three-deep ring)
  if (l.in_bf16 && l.mode == kVecLoads && l.to == 128 && l.big_tile_ok) {
    long pix = 0;
    for (int c = 0; c < p.tt.n_classes; c++) pix += (long)p.N * p.tt.cls_oh[c] * p.tt.cls_ow[c];
    static const long big_min = getenv("GC_BIG_TILE_MIN") ? atol(getenv("GC_BIG_TILE_MIN")) : 1024;
    if ((pix / 256) * p.n_oc_tiles >= big_min) l.tp = 256;
    // largest problems: 256 x 256 per workgroup (two-deep ring fills the LDS) halves the bytes moved into LDS per FLOP
    if (l.tp == 256 && p.OC % 256 == 0 && (pix / 256) * (p.OC / 256) >= 512) {
      l.to = 256;
      p.n_oc_tiles = p.OC / 256;
    }
  }
  int tiles = 0, min_steps = 1 << 30;
  for (int c = 0; c < p.tt.n_classes; c++) {
    p.tt.cls_tile_start[c] = tiles;
    const long mc = (long)p.N * p.tt.cls_oh[c] * p.tt.cls_ow[c];
    tiles += (int)((mc + l.tp - 1) / l.tp);
    const int steps = l.mode == kThinInput ? (p.T * p.ICp + kBK - 1) / kBK
                                           : (p.tt.cls_tap_start[c + 1] - p.tt.cls_tap_start[c]) * (p.ICp / kBK);
    if (steps < min_steps) min_steps = steps;
  }
  for (int c = p.tt.n_classes; c <= kMaxClasses; c++) p.tt.cls_tile_start[c] = tiles;
  p.n_pix_tiles = tiles;
  p.cls_interleave = 0;
  if (p.tt.n_classes > 1) {
    p.cls_interleave = 1;
    const int per = p.tt.cls_tile_start[1] - p.tt.cls_tile_start[0];
    for (int c = 1; c < p.tt.n_classes; c++)
      if (p.tt.cls_tile_start[c + 1] - p.tt.cls_tile_start[c] != per) p.cls_interleave = 0;
  }
  // Split the reduction when the grid cannot fill the chip and the output can take fp32 adds.
  p.ksplit = 1;
  const long blocks = (long)tiles * p.n_oc_tiles;
  const bool can_split = split_reduction_allowed() && p.out_f32 != 0 && p.act == GC_ACT_NONE && p.ch_sum == nullptr && p.stat_rows == nullptr && p.out_bf16 == nullptr && p.out_cstride == p.OC && p.out_coff == 0 &&
                         p.pre_scale == nullptr && p.post_scale == nullptr;
  // (only for small outputs: the partial tiles are added with per-lane fp32 atomics whose access shape is slow)
  static const long split_max_out = getenv("GC_SPLIT_MAX_OUT") ? atol(getenv("GC_SPLIT_MAX_OUT")) : 65536;
  if (can_split && blocks > 0 && blocks < 128 && min_steps >= 16 && (long)p.N * p.OH * p.OW * p.OC <= split_max_out) {
    long want = (256 + blocks - 1) / blocks;
    if (want > min_steps / 8) want = min_steps / 8;
    if (want > 32) want = 32;
    if (want > 1) p.ksplit = (int)want;
  }
}

template <typename TIn, int TO, int TP, bool X3, int MODE>
static int launch_one(const GatherParams& p, double flops, hipStream_t stream) {
  constexpr int PL = X3 ? 2 : 1;
  constexpr int smem = 2 * PL * (TO + TP) * 128 + kMaxTaps * 4;
  auto kern = conv_gather_kernel<TIn, TO, TP, X3, MODE>;
  if (smem > 64 * 1024 && ensure_dynamic_smem(reinterpret_cast<const void*>(kern), smem) != GC_OK) return GC_HIP_ERROR;
  if (p.ksplit > 1) {
    const size_t bytes = (size_t)p.N * p.OH * p.OW * p.OC * sizeof(float);
    if (zero_fill(p.out, bytes, stream) != GC_OK) return GC_HIP_ERROR;
  }
  const unsigned nblocks = (unsigned)p.n_pix_tiles * (unsigned)p.n_oc_tiles * (unsigned)p.ksplit;
  char name[96];  // the demangled kernel name as a profiler prints it
  snprintf(name, sizeof(name), "gc::conv_gather_kernel<%s, %d, %d, %s, %d>", sizeof(TIn) == 4 ? "float" : "unsigned short", TO, TP,
           X3 ? "true" : "false", MODE);
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kThreads), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_gather launch");
}

template <typename TIn, bool X3, int MODE>
static int launch_tile(const GatherParams& p, const GatherLaunch& l, hipStream_t s) {
  if (l.to == 128) return launch_one<TIn, 128, 128, X3, MODE>(p, l.flops, s);
  if (l.to == 64) return launch_one<TIn, 64, 128, X3, MODE>(p, l.flops, s);
  return launch_one<TIn, 16, 256, X3, MODE>(p, l.flops, s);
}

template <typename TIn, bool X3>
static int launch_mode(const GatherParams& p, const GatherLaunch& l, hipStream_t s) {
  if (l.mode == kVecLoads) return launch_tile<TIn, X3, kVecLoads>(p, l, s);
  if (l.mode == kElemLoads) return launch_tile<TIn, X3, kElemLoads>(p, l, s);
  return launch_tile<TIn, X3, kThinInput>(p, l, s);
}

int launch_gather(const GatherParams& p0, const GatherLaunch& l, hipStream_t s) {
  if (p0.n_pix_tiles == 0 || p0.n_oc_tiles == 0) return GC_OK;
  GatherParams p = p0;
  if (p.stat_rows != nullptr) {     // one row per wave's pixel share of a tile; no rows when they do not fit the caller's table
    const int rows = p.n_pix_tiles * (l.to >= 64 ? 2 : 4);
    if (p.ksplit == 1 && rows <= p.stat_cap) *p.stat_written = rows;
    else p.stat_rows = nullptr;
  }
  if (l.in_bf16) return launch_mode<uint16_t, false>(p, l, s);
  return l.x3 ? launch_mode<float, true>(p, l, s) : launch_mode<float, false>(p, l, s);
}

}  // namespace gc
