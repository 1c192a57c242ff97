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
#include <stdio.h>

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

template <typename TIn, int TO, int TP, bool X3, bool ALIGNED>
__global__ __launch_bounds__(kThreads) void conv_gather_kernel(const GatherParams p) {
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

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int wo = wave / C::WP, wp_ = wave % C::WP;

  // ---- which tile ------------------------------------------------------------------------------------
  const uint32_t nblocks = (uint32_t)p.n_pix_tiles * (uint32_t)p.n_oc_tiles;
  const uint32_t lid = xcd_remap(blockIdx.x, nblocks);
  const int pix_tile = lid / p.n_oc_tiles;
  const int oc0 = (lid % p.n_oc_tiles) * TO;
  int cls = 0;
#pragma unroll
  for (int c = 1; c < kMaxClasses; c++)
    if (c < p.tt.n_classes && pix_tile >= p.tt.cls_tile_start[c]) cls = c;
  const int m0 = (pix_tile - p.tt.cls_tile_start[cls]) * TP;
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

  u32x4 areg[AR][AC];
  u32x4 wreg[PL][WR];

  auto load_tile = [&](int tap, int c0) {
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
        areg[i][j] = v;
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
            v = *reinterpret_cast<const u32x4*>(p.wp + pl * p.plane_stride + ((size_t)oc * p.T + wt) * p.ICp + c0 + (t & 7) * 8);
          wreg[pl][i] = v;
        }
    }
  };

  auto store_tile = [&](int buf) {
    char* wb = smem + buf * STAGE;
    char* ab = wb + PL * W_BYTES;
#pragma unroll
    for (int i = 0; i < AR; i++) {
      const int row = (t >> 2) + 64 * i;
#pragma unroll
      for (int j = 0; j < AC; j++) {
        const int ch = (t & 3) + 4 * j;
        if (sizeof(TIn) == 4) {
          const f32x4 v = __builtin_bit_cast(f32x4, areg[i][j]);
          const u32x2 hi = cvt_bf16x4(v);
          const int off = row * 128 + (((ch >> 1) ^ swz(row)) << 4) + (ch & 1) * 8;
          *reinterpret_cast<u32x2*>(ab + off) = hi;
          if (X3) *reinterpret_cast<u32x2*>(ab + A_BYTES + off) = cvt_bf16x4(v - widen_bf16x4(hi));
        } else {
          *reinterpret_cast<u32x4*>(ab + row * 128 + ((ch ^ swz(row)) << 4)) = areg[i][j];
        }
      }
    }
    if (TO >= 32 || t < TO * 8) {
#pragma unroll
      for (int pl = 0; pl < PL; pl++)
#pragma unroll
        for (int i = 0; i < WR; i++) {
          const int row = (t >> 3) + 32 * i;
          *reinterpret_cast<u32x4*>(wb + pl * W_BYTES + row * 128 + (((t & 7) ^ swz(row)) << 4)) = wreg[pl][i];
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
  const int csteps = p.ICp / kBK;
  const int nsteps = (tap_end - tap_begin) * csteps;
  int tap = tap_begin, cs = 0;
  if (nsteps > 0) {
    load_tile(tap, 0);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < nsteps; it++) {
    const bool more = it + 1 < nsteps;
    if (more) {
      if (++cs == csteps) { cs = 0; tap++; }
      load_tile(tap, cs * kBK);
    }
    compute(it & 1);
    if (more) store_tile((it + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue ----------------------------------------------------------------------------------------
  const bool vec_ok = p.out_vec != 0;
  float ssum[C::FO][4], ssq[C::FO][4];
#pragma unroll
  for (int i = 0; i < C::FO; i++)
#pragma unroll
    for (int r = 0; r < 4; r++) ssum[i][r] = ssq[i][r] = 0.f;
  const bool stats = p.ch_sum != nullptr;
  TIn* out = reinterpret_cast<TIn*>(p.out);
#pragma unroll
  for (int j = 0; j < C::FP; j++) {
    const int m = m0 + wp_ * (TP / C::WP) + j * 16 + (lane & 15);
    const bool mok = m < Mc;
    const int mm = mok ? m : 0;
    const int ow = mm % OWc, tmp = mm / OWc;
    const int oh = tmp % OHc, n = tmp / OHc;
    const size_t pix = ((size_t)n * p.OH + (oh * p.b + p.tt.cls_ph[cls])) * p.OW + (ow * p.b + p.tt.cls_pw[cls]);
    TIn* dst = out + pix * p.out_cstride + p.out_coff;
#pragma unroll
    for (int i = 0; i < C::FO; i++) {
      const int oc = oc0 + wo * (TO / C::WO) + i * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float x = acc[i][j][r];
        if (p.bias != nullptr && oc + r < p.OC) x += p.bias[oc + r];
        x = apply_act(x, p.act, p.slope);
        if (sizeof(TIn) == 2) x = bf16_bits_to_float(float_to_bf16_bits(x));
        v[r] = x;
        if (stats && mok && oc + r < p.OC) { ssum[i][r] += x; ssq[i][r] += x * x; }
      }
      if (!mok) continue;
      if (vec_ok && oc + 3 < p.OC) {
        if (sizeof(TIn) == 4) {
          *reinterpret_cast<f32x4*>(dst + oc) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
          *reinterpret_cast<u32x2*>(dst + oc) = cvt_bf16x4(f32x4{v[0], v[1], v[2], v[3]});
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oc + r < p.OC) {
            if (sizeof(TIn) == 4) reinterpret_cast<float*>(dst)[oc + r] = v[r];
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
          atomicAdd(p.ch_sum + oc, s);
          if (p.ch_sumsq != nullptr) atomicAdd(p.ch_sumsq + oc, q);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------
void plan_gather(GatherParams& p, GatherLaunch& l) {
  // Channel tile: the smallest of {16, 64, 128} covering OC (128 when OC > 64); pixel tile 128, or 256 for the
  // 16-channel tile (thin outputs such as RGB images or critic scores).
  l.to = p.OC <= 16 ? 16 : (p.OC <= 64 ? 64 : 128);
  l.tp = l.to == 16 ? 256 : 128;
  p.n_oc_tiles = (p.OC + l.to - 1) / l.to;
  int tiles = 0;
  for (int c = 0; c < p.tt.n_classes; c++) {
    p.tt.cls_tile_start[c] = tiles;
    const long mc = (long)p.N * p.tt.cls_oh[c] * p.tt.cls_ow[c];
    tiles += (int)((mc + l.tp - 1) / l.tp);
  }
  for (int c = p.tt.n_classes; c <= kMaxClasses; c++) p.tt.cls_tile_start[c] = tiles;
  p.n_pix_tiles = tiles;
}

template <typename TIn, int TO, int TP, bool X3, bool ALIGNED>
static int launch_one(const GatherParams& p, double flops, hipStream_t stream) {
  constexpr int PL = X3 ? 2 : 1;
  constexpr int smem = 2 * PL * (TO + TP) * 128;
  auto kern = conv_gather_kernel<TIn, TO, TP, X3, ALIGNED>;
  static bool attr_set = false;
  if (smem > 64 * 1024 && !attr_set) {
    if (check_hip(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem),
                  "hipFuncSetAttribute"))
      return GC_HIP_ERROR;
    attr_set = true;
  }
  const unsigned nblocks = (unsigned)p.n_pix_tiles * (unsigned)p.n_oc_tiles;
  char name[64];  // the demangled kernel name as a profiler prints it
  snprintf(name, sizeof(name), "gc::conv_gather_kernel<%s, %d, %d, %s, %s>", sizeof(TIn) == 4 ? "float" : "unsigned short", TO, TP,
           X3 ? "true" : "false", ALIGNED ? "true" : "false");
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(kThreads), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_gather launch");
}

template <typename TIn, bool X3, bool ALIGNED>
static int launch_tile(const GatherParams& p, const GatherLaunch& l, hipStream_t s) {
  if (l.to == 128) return launch_one<TIn, 128, 128, X3, ALIGNED>(p, l.flops, s);
  if (l.to == 64) return launch_one<TIn, 64, 128, X3, ALIGNED>(p, l.flops, s);
  return launch_one<TIn, 16, 256, X3, ALIGNED>(p, l.flops, s);
}

int launch_gather(const GatherParams& p, const GatherLaunch& l, hipStream_t s) {
  if (p.n_pix_tiles == 0 || p.n_oc_tiles == 0) return GC_OK;
  if (l.in_bf16) {
    return l.aligned ? launch_tile<uint16_t, false, true>(p, l, s) : launch_tile<uint16_t, false, false>(p, l, s);
  }
  if (l.x3) return l.aligned ? launch_tile<float, true, true>(p, l, s) : launch_tile<float, true, false>(p, l, s);
  return l.aligned ? launch_tile<float, false, true>(p, l, s) : launch_tile<float, false, false>(p, l, s);
}

}  // namespace gc
