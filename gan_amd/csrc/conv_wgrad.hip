// Weight gradient on the bf16 matrix cores:  gw[tap][co][ci] = sum over pixels k of  gy[k][co] * x[gather(k, tap)][ci]
//
// GEMM view: rows = output channels, columns = input channels (of one filter tap), reduction = the N*Ho*Wo
// output pixels.  Both operands are stored pixel-major in HBM (NHWC), i.e. with the reduction index on the slow
// axis, so the LDS tiles keep that orientation ([pixel][channel], XOR-swizzled 32-byte segments) and the MFMA
// fragments are formed with the transposing LDS read (ds_read_b64_tr_b16).  The reduction is split across
// workgroups (grid.z); every split writes its partial tile with plain stores and a second small kernel sums
// the splits in a fixed order, scales, and scatters into the caller's [Cout][Cin][kh][kw] layout -- bitwise
// reproducible, no float atomics.
#include <stdio.h>
#include <stdlib.h>

#include "conv_wgrad.h"

namespace gc {

constexpr int kWBK = 32;  // pixels per reduction step

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int ROWB>
__device__ __forceinline__ int wswz(int k) {
  return ROWB == 256 ? ((k & 3) | (((k >> 3) & 1) << 2)) : (((k >> 1) & 1) | (((k >> 3) & 1) << 1));
}

// THIN: up to 32 input channels (RGB images ...): one tile's columns cover (tap, channel) pairs, so a 3-channel 4x4
// filter is one 64-column tile instead of sixteen 97 %-empty ones.
template <typename T, int TO, int TC, bool X3, bool ALIGNED, bool THIN>
__global__ __launch_bounds__(kThreads) void conv_wgrad_kernel(const WgradParams p) {
  constexpr int PL = X3 ? 2 : 1;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int CPRY = TO / EPC, RPPY = kThreads / CPRY, PY = kWBK / RPPY;
  constexpr int CPRX = TC / EPC, RPPX = kThreads / CPRX, PX = kWBK / RPPX;
  constexpr int ROWBY = TO * 2, ROWBX = TC * 2;
  constexpr int Y_BYTES = kWBK * ROWBY, X_BYTES = kWBK * ROWBX;
  constexpr int STAGE = PL * (Y_BYTES + X_BYTES);
  constexpr int FO = TO / 32, FC = TC / 32;
  static_assert(PY >= 1 && PX >= 1, "tile too narrow for the load mapping");

  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wo = wave >> 1, wc = wave & 1;
  // Logical order (tile, tap) fastest, split slowest, dealt to the XCDs in contiguous runs: the workgroups of one
  // split read the same pixels of gy and x, so they should meet in one L2.
  const uint32_t ntiles = (uint32_t)p.n_co_tiles * (uint32_t)p.n_ci_tiles, ntaps = THIN ? 1u : (uint32_t)(p.kh * p.kw);
  uint32_t lid = xcd_remap(blockIdx.x, ntiles * ntaps * (uint32_t)p.splits);
  const int tile = (int)(lid % ntiles);
  lid /= ntiles;
  const int tap = (int)(lid % ntaps), r = tap / p.kw, s = tap % p.kw;
  const int split = (int)(lid / ntaps);
  const int co0 = (tile / p.n_ci_tiles) * TO, ci0 = (tile % p.n_ci_tiles) * TC;
  const WgradRange rg = wgrad_range(p, split);
  const int kb0 = rg.kb0, kend = rg.kend;

  const T* gy = reinterpret_cast<const T*>(rg.gy);
  const T* x = reinterpret_cast<const T*>(rg.x);
  const int cy = t % CPRY, cx = t % CPRX;

  u32x4 yreg[PY], xreg[PX];

  // thin mode: this thread's chunk of columns is fixed, so the (tap, channel) of each group of 4 columns is too
  int thin_r[EPC / 4], thin_s[EPC / 4], thin_c[EPC / 4];
  if (THIN) {
#pragma unroll
    for (int sub = 0; sub < EPC / 4; sub++) {
      const int col = ci0 + cx * EPC + sub * 4;
      const int tp = col >> p.icp_log2;
      thin_c[sub] = col & (p.ICp - 1);
      thin_r[sub] = tp < p.kh * p.kw ? tp / p.kw : -100000;   // out of range -> never valid
      thin_s[sub] = tp % p.kw;
    }
  }

  auto load_chunk = [&](const T* src, int c_first, int c_limit, bool ok, bool vec) -> u32x4 {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ALIGNED && vec) {
      if (ok && c_first < c_limit) v = *reinterpret_cast<const u32x4*>(src);
    } else if (ok) {
      if (sizeof(T) == 4) {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (c_first + e < c_limit) v[e] = reinterpret_cast<const uint32_t*>(src)[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++)
          if (c_first + e < c_limit) v[e >> 1] |= (uint32_t)reinterpret_cast<const uint16_t*>(src)[e] << (16 * (e & 1));
      }
    }
    return v;
  };

  auto load_tile = [&](int kb) {
#pragma unroll
    for (int i = 0; i < PY; i++) {
      const int k = kb + t / CPRY + RPPY * i;
      const int c = co0 + cy * EPC;
      yreg[i] = load_chunk(gy + ((size_t)k * p.y_cstride + p.y_coff + c), c, p.Cout, k < kend, p.y_elem == 0);
    }
#pragma unroll
    for (int i = 0; i < PX; i++) {
      const int k = kb + t / CPRX + RPPX * i;
      const uint32_t kk = k < kend ? (uint32_t)k : 0u;
      const uint32_t q1 = fd_div(kk, p.div_wo);
      const int wo_ = (int)(kk - q1 * p.div_wo.d);
      const uint32_t n = fd_div(q1, p.div_ho);
      const int ho_ = (int)(q1 - n * p.div_ho.d);
      if (THIN) {
        u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int sub = 0; sub < EPC / 4; sub++) {
          const int ih = ho_ * p.sh - p.ph + thin_r[sub], iw = wo_ * p.sw - p.pw + thin_s[sub];
          if (k < kend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) {
            const T* src = x + (((size_t)n * p.H + ih) * p.W + iw) * p.x_cstride + p.x_coff + thin_c[sub];
#pragma unroll
            for (int q = 0; q < 4; q++)
              if (thin_c[sub] + q < p.Cin) {
                if (sizeof(T) == 4) v[q] = reinterpret_cast<const uint32_t*>(src)[q];
                else v[sub * 2 + (q >> 1)] |= (uint32_t)reinterpret_cast<const uint16_t*>(src)[q] << (16 * (q & 1));
              }
          }
        }
        xreg[i] = v;
        continue;
      }
      const int ih = ho_ * p.sh - p.ph + r, iw = wo_ * p.sw - p.pw + s;
      const bool ok = k < kend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const int c = ci0 + cx * EPC;
      xreg[i] = load_chunk(x + (((size_t)n * p.H + ih) * p.W + iw) * p.x_cstride + p.x_coff + c, c, p.Cin, ok, true);
    }
  };

  auto store_one = [&](char* base, int plane_bytes, int rowb_sel, int row, int ch, u32x4 v) {
    // ch: 16-byte chunk index of the source row; rowb_sel: 0 -> gy tile, 1 -> x tile
    const int rowb = rowb_sel ? ROWBX : ROWBY;
    const int f = rowb_sel ? wswz<ROWBX>(row) : wswz<ROWBY>(row);
    if (sizeof(T) == 4) {
      const f32x4 fv = __builtin_bit_cast(f32x4, v);
      const u32x2 hi = cvt_bf16x4(fv);
      const int off = row * rowb + (((ch >> 2) ^ f) << 5) + (ch & 3) * 8;
      *reinterpret_cast<u32x2*>(base + off) = hi;
      if (X3) *reinterpret_cast<u32x2*>(base + plane_bytes + off) = cvt_bf16x4(fv - widen_bf16x4(hi));
    } else {
      *reinterpret_cast<u32x4*>(base + row * rowb + (((ch >> 1) ^ f) << 5) + (ch & 1) * 16) = v;
    }
  };

  auto store_tile = [&](int buf) {
    char* yb = smem + buf * STAGE;
    char* xb = yb + PL * Y_BYTES;
#pragma unroll
    for (int i = 0; i < PY; i++) store_one(yb, Y_BYTES, 0, t / CPRY + RPPY * i, cy, yreg[i]);
#pragma unroll
    for (int i = 0; i < PX; i++) store_one(xb, X_BYTES, 1, t / CPRX + RPPX * i, cx, xreg[i]);
  };

  f32x4 acc[FO][FC];
#pragma unroll
  for (int i = 0; i < FO; i++)
#pragma unroll
    for (int j = 0; j < FC; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposing read: lane (g = lane>>4, q = (lane>>2)&3, pp = lane&3) addresses row 8g+q (+4), columns 4pp..4pp+3
  // of a 16-channel segment and receives, for channel lane&15, the four rows 8g..8g+3 (+4): exactly k = 8g+j.
  const int k0 = 8 * (lane >> 4) + ((lane >> 2) & 3);
  const int sub = (lane & 3) * 8;

  auto read_frag = [&](const char* base, int rowb, int f, int seg) -> bf16x8 {
    const char* a0 = base + k0 * rowb + ((seg ^ f) << 5) + sub;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 4 * rowb));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  auto compute = [&](int buf) {
    const char* yb = smem + buf * STAGE;
    const char* xb = yb + PL * Y_BYTES;
    const int fy = wswz<ROWBY>(k0), fx = wswz<ROWBX>(k0);
    bf16x8 a[PL][FO], b[PL][FC];
#pragma unroll
    for (int pl = 0; pl < PL; pl++) {
#pragma unroll
      for (int i = 0; i < FO; i++) a[pl][i] = read_frag(yb + pl * Y_BYTES, ROWBY, fy, wo * FO + i);
#pragma unroll
      for (int j = 0; j < FC; j++) b[pl][j] = read_frag(xb + pl * X_BYTES, ROWBX, fx, wc * FC + j);
    }
#pragma unroll
    for (int i = 0; i < FO; i++)
#pragma unroll
      for (int j = 0; j < FC; j++) {
        if (X3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1][i], b[0][j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[1][j], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0][i], b[0][j], acc[i][j], 0, 0, 0);
      }
  };

  const int nsteps = kend > kb0 ? (kend - kb0 + kWBK - 1) / kWBK : 0;
  if (nsteps > 0) {
    load_tile(kb0);
    store_tile(0);
  }
  __syncthreads();
  for (int it = 0; it < nsteps; it++) {
    const bool more = it + 1 < nsteps;
    if (more) load_tile(kb0 + (it + 1) * kWBK);
    compute(it & 1);
    if (more) store_tile((it + 1) & 1);
    __syncthreads();
  }

  const int ncols = THIN ? p.ncols : p.Cin;
  float* part = THIN ? p.part + (size_t)split * p.Cout * ncols : p.part + ((size_t)split * (p.kh * p.kw) + tap) * p.Cout * p.Cin;
#pragma unroll
  for (int i = 0; i < FO; i++)
#pragma unroll
    for (int j = 0; j < FC; j++) {
      const int ci = ci0 + wc * (TC / 2) + j * 16 + (lane & 15);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int co = co0 + wo * (TO / 2) + i * 16 + (lane >> 4) * 4 + e;
        if (co < p.Cout && ci < ncols) part[(size_t)co * ncols + ci] = acc[i][j][e];
      }
    }
}

// Sums the split-K partials in a fixed order, scales and scatters into the caller's weight-gradient layout.
// Workgroup = 64 runs of V consecutive partial elements x SL split lanes (SL = waves of the launch: 4, or 16 for the
// many short slabs of a thin layer, whose few output elements would otherwise each wait for a long serial sum); each lane
// keeps four loads in flight.
// The workgroups behind the main ones add the per-split bias partials [splits][Cout] in split order: 64 channels per
// workgroup, wave w takes the splits w, w + SL, ..., wave 0 adds the waves' sums in order.
__device__ __forceinline__ void wgrad_bias_reduce(const WgradReduceParams& p, float* red) {
  const int SL = blockDim.x >> 6, lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int co = (int)(blockIdx.x - p.main_blocks) * 64 + lane;
  float sum = 0.f;
  if (co < p.Cout)
    for (int sp = sl; sp < p.splits; sp += SL) sum += p.bias_part[(size_t)sp * p.Cout + co];
  red[threadIdx.x] = sum;
  __syncthreads();
  if (sl != 0 || co >= p.Cout) return;
  for (int w = 1; w < SL; w++) sum += red[lane + 64 * w];
  p.gbias[co] = sum;
}

template <int V>
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const WgradReduceParams p) {
  typedef float vec_t __attribute__((ext_vector_type(V)));
  __shared__ vec_t red[1024];
  if (p.bias_part != nullptr && (int)blockIdx.x >= p.main_blocks) { wgrad_bias_reduce(p, reinterpret_cast<float*>(red)); return; }
  const int SL = blockDim.x >> 6;
  const int T = p.kh * p.kw;
  const int inner = p.thin ? p.ICp : p.Cin;
  const size_t total = (size_t)T * p.Cout * inner;      // elements of one partial slab
  const size_t e0 = ((size_t)blockIdx.x * 64 + (threadIdx.x & 63)) * V;
  const int sl = threadIdx.x >> 6;
  vec_t sum = 0.f;
  if (e0 < total) {
    const float* src = p.part + e0;
    int sp = sl;
    for (; sp + 3 * SL < p.splits; sp += 4 * SL) {
      const vec_t v0 = *reinterpret_cast<const vec_t*>(src + (size_t)sp * total);
      const vec_t v1 = *reinterpret_cast<const vec_t*>(src + (size_t)(sp + SL) * total);
      const vec_t v2 = *reinterpret_cast<const vec_t*>(src + (size_t)(sp + 2 * SL) * total);
      const vec_t v3 = *reinterpret_cast<const vec_t*>(src + (size_t)(sp + 3 * SL) * total);
      sum += (v0 + v1) + (v2 + v3);
    }
    for (; sp < p.splits; sp += SL) sum += *reinterpret_cast<const vec_t*>(src + (size_t)sp * total);
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  if (sl != 0 || e0 >= total) return;
  sum = red[threadIdx.x];
  for (int w = 1; w < SL; w++) sum += red[threadIdx.x + 64 * w];      // fixed order
#pragma unroll
  for (int q = 0; q < V; q++) {
    const size_t e = e0 + q;
    int co, ci, tap;
    if (p.thin) {          // [Cout][tap][ICp]
      ci = (int)(e % p.ICp);
      tap = (int)((e / p.ICp) % T);
      co = (int)(e / ((size_t)p.ICp * T));
      if (ci >= p.Cin) continue;
    } else {               // [tap][Cout][Cin]
      ci = (int)(e % p.Cin);
      co = (int)((e / p.Cin) % p.Cout);
      tap = (int)(e / ((size_t)p.Cin * p.Cout));
    }
    float* dst = p.gw + co * p.s_co + ci * p.s_ci + (tap / p.kw) * p.s_r + (tap % p.kw) * p.s_s;
    const float v = p.alpha * sum[q];
    *dst = p.accumulate ? *dst + v : v;
  }
}

// The same reduction for the common case of a dense [Cout][Cin][kh][kw] gradient with at most 16 taps.  A workgroup
// owns 64 consecutive (co, ci) pairs for all taps; each of its SW waves takes every SW-th slab and reads, per
// instruction, four taps x 256 contiguous bytes (lane = (tap % 4, pair quad)).  Waves 1.. hand their sums to wave 0
// through LDS, which adds them in a fixed order, transposes [tap][pair] -> [pair][tap] through LDS and writes the
// workgroup's 64 * T results as one contiguous run (the generic kernel above writes 4-byte pieces 4 T bytes apart).
template <int SW>
__global__ __launch_bounds__(SW * 64) void wgrad_reduce_dense_kernel(const WgradReduceParams p) {
  __shared__ f32x4 red[(SW > 2 ? SW - 1 : 2) * 64 * 4];   // [wave - 1][j][lane]; reused as the [pair][tap] tile (4352 bytes)
  if (p.bias_part != nullptr && (int)blockIdx.x >= p.main_blocks) { wgrad_bias_reduce(p, reinterpret_cast<float*>(red)); return; }
  const int T = p.kh * p.kw;
  const size_t pairs = (size_t)p.Cout * p.Cin;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quad = lane & 15, tl = lane >> 4;
  const size_t pair0 = (size_t)blockIdx.x * 64;
  const bool ok = pair0 + quad * 4 < pairs;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (ok) {
    // two slabs per step: eight loads in flight per lane (a slab at a time leaves one memory latency per slab in a row);
    // the order of the additions is that of the slab index either way
    int sp = wave;
    for (; sp + SW < p.splits; sp += 2 * SW) {
      const float* src = p.part + (size_t)sp * T * pairs + pair0 + quad * 4;
      const float* src2 = src + (size_t)SW * T * pairs;
      f32x4 a[4], b[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        a[j] = b[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tl + 4 * j < T) {
          a[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(tl + 4 * j) * pairs);
          b[j] = *reinterpret_cast<const f32x4*>(src2 + (size_t)(tl + 4 * j) * pairs);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) { acc[j] += a[j]; acc[j] += b[j]; }
    }
    for (; sp < p.splits; sp += SW) {
      const float* src = p.part + (size_t)sp * T * pairs + pair0 + quad * 4;
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (tl + 4 * j < T) acc[j] += *reinterpret_cast<const f32x4*>(src + (size_t)(tl + 4 * j) * pairs);
    }
  }
  if (SW > 1) {
    if (wave > 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) red[((wave - 1) * 4 + j) * 64 + lane] = acc[j];
    }
    __syncthreads();
    if (wave > 0) return;
    for (int w = 0; w < SW - 1; w++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] += red[(w * 4 + j) * 64 + lane];
  }
  // from here on one wave works alone; the LDS unit serves its instructions in order, so the tile needs no barrier
  float* tile = reinterpret_cast<float*>(red);   // [64 pairs][17]: the odd pitch keeps the column writes off one bank
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) tile[(quad * 4 + e) * 17 + tl + 4 * j] = acc[j][e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // lane t of the run: element (pair0 * T + t), i.e. pair t / T, tap t % T -- contiguous in the gradient
  const int n_out = (int)((pairs - pair0 < 64 ? pairs - pair0 : 64) * T);
  float* dst = p.gw + pair0 * T;
  for (int t = lane; t < n_out; t += 64) {
    const float v = p.alpha * tile[(t / T) * 17 + t % T];
    dst[t] = p.accumulate ? dst[t] + v : v;
  }
}

// gb[c] += sum over pixels of gy[k][c]; gb is zeroed by the launcher.  One block sums a run of pixels.
template <typename T>
__global__ __launch_bounds__(kThreads) void colsum_kernel(const T* gy, float* gb, int K, int C, int cstride, int coff, int k_per_block) {
  const int k0 = blockIdx.x * k_per_block, k1 = min(K, k0 + k_per_block);
  // threads are laid out [pixel lane][channel]: tc channels wide
  const int tc = C >= kThreads ? kThreads : C;
  const int tk = kThreads / tc;
  const int c_lane = threadIdx.x % tc, k_lane = threadIdx.x / tc;
  __shared__ float red[kThreads];
  for (int cb = 0; cb < C; cb += tc) {
    const int c = cb + c_lane;
    float s = 0.f;
    if (k_lane < tk && c < C)
      for (int k = k0 + k_lane; k < k1; k += tk) {
        const T v = gy[(size_t)k * cstride + coff + c];
        s += sizeof(T) == 4 ? __builtin_bit_cast(float, (uint32_t)v) : bf16_bits_to_float((uint16_t)v);
      }
    red[threadIdx.x] = s;
    __syncthreads();
    if (k_lane == 0 && c < C) {
      for (int j = 1; j < tk; j++) s += red[j * tc + c_lane];
      atomicAdd(gb + c, s);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
void plan_wgrad(WgradParams& p, WgradLaunch& l, bool allow_dma) {
  l.to = p.Cout <= 64 ? 64 : 128;
  p.thin = p.Cin <= 32;
  p.image = 0;
  p.n_tap_groups = p.kh * p.kw; p.tpt = 1; p.cw_log2 = 7;
  l.dma = allow_dma && l.bf16 && !l.x3 && !p.thin && p.Cin % 8 == 0 && p.Cout % 8 == 0 && p.x_cstride % 8 == 0 &&
          p.x_coff % 8 == 0 && p.y_cstride % 8 == 0 && p.y_coff % 8 == 0;
  p.x_bytes = (int64_t)p.N * p.H * p.W * p.x_cstride * 2;
  p.y_bytes = (int64_t)p.K * p.y_cstride * 2;
  const int64_t limit = (1ll << 31) - (1 << 20);   // 32-bit buffer offsets
  if (p.x_bytes >= limit || p.y_bytes >= limit) l.dma = false;
  if (l.dma) {
    l.tc = 128;
    if (p.Cin <= 64) {   // two taps of up to 64 channels share one tile (and its gy panel)
      p.cw_log2 = 6; p.tpt = 2; p.n_ci_tiles = 1; p.n_tap_groups = (p.kh * p.kw + 1) / 2;
    } else {
      p.n_ci_tiles = (p.Cin + 127) / 128;
    }
    // 256 output channels per tile (fewer bytes into LDS per FLOP) where that does not multiply the partial sums:
    // long reductions, or weights with enough tiles to fill the chip anyway
    static const int force256 = getenv("GC_WGRAD_TO256") ? atoi(getenv("GC_WGRAD_TO256")) : 0;
    if (p.Cout % 256 == 0 && (force256 || p.K >= 65536 || (long)(p.Cout / 256) * p.n_ci_tiles * p.n_tap_groups >= 256)) l.to = 256;
  } else if (p.thin) {
    p.image = wgrad_image_geometry(p) ? 1 : 0;
    p.ICp = 4;
    while (p.ICp < p.Cin) p.ICp *= 2;
    for (p.icp_log2 = 0; (1 << p.icp_log2) < p.ICp; p.icp_log2++) {}
    p.ncols = p.kh * p.kw * p.ICp;
    l.tc = p.ncols <= 64 ? 64 : 128;
    p.n_ci_tiles = (p.ncols + l.tc - 1) / l.tc;
  } else {
    l.tc = p.Cin <= 64 ? 64 : 128;
    p.n_ci_tiles = (p.Cin + l.tc - 1) / l.tc;
  }
  p.n_co_tiles = (p.Cout + l.to - 1) / l.to;
  const long tiles = (long)p.n_co_tiles * p.n_ci_tiles * (p.thin ? 1 : p.n_tap_groups);
  if (p.K <= 0) { p.k_per_split = kWBK; p.splits = 1; return; }
  // enough splits for about two workgroups per CU, each split at least 8 reduction steps long
  static const int wg_target = getenv("GC_WGRAD_WGS") ? atoi(getenv("GC_WGRAD_WGS")) : 512;
  long want = (wg_target + tiles - 1) / tiles;
  const long max_splits = (p.K + 8 * kWBK - 1) / (8 * kWBK);
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  static const int max_want = getenv("GC_WGRAD_MAX_SPLITS") ? atoi(getenv("GC_WGRAD_MAX_SPLITS")) : 512;
  if (want > max_want) want = max_want;
  long kps = (p.K + want - 1) / want;
  const int step = p.image ? 128 : kWBK;
  kps = (kps + step - 1) / step * step;
  p.k_per_split = (int)kps;
  p.splits = (int)((p.K + kps - 1) / kps);
}

template <typename T, int TO, int TC, bool X3, bool ALIGNED, bool THIN>
static int launch_one(const WgradParams& p, double flops, hipStream_t stream) {
  char name[96];
  snprintf(name, sizeof(name), "gc::conv_wgrad_kernel<%s, %d, %d, %s, %s, %s>", sizeof(T) == 4 ? "float" : "unsigned short", TO, TC,
           X3 ? "true" : "false", ALIGNED ? "true" : "false", THIN ? "true" : "false");
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL((conv_wgrad_kernel<T, TO, TC, X3, ALIGNED, THIN>),
                     dim3((unsigned)(p.n_co_tiles * p.n_ci_tiles * (THIN ? 1 : p.kh * p.kw) * p.splits)), dim3(kThreads), 0, stream, p);
  return check_hip(hipGetLastError(), "conv_wgrad launch");
}

template <typename T, bool X3, bool ALIGNED>
static int launch_tile(const WgradParams& p, const WgradLaunch& l, hipStream_t s) {
  if (p.thin) {   // the gy side decides ALIGNED here; the thin x side always uses element loads
    if (l.to == 128 && l.tc == 128) return launch_one<T, 128, 128, X3, ALIGNED, true>(p, l.flops, s);
    if (l.to == 128 && l.tc == 64) return launch_one<T, 128, 64, X3, ALIGNED, true>(p, l.flops, s);
    if (l.to == 64 && l.tc == 128) return launch_one<T, 64, 128, X3, ALIGNED, true>(p, l.flops, s);
    return launch_one<T, 64, 64, X3, ALIGNED, true>(p, l.flops, s);
  }
  if (l.to == 128 && l.tc == 128) return launch_one<T, 128, 128, X3, ALIGNED, false>(p, l.flops, s);
  if (l.to == 128 && l.tc == 64) return launch_one<T, 128, 64, X3, ALIGNED, false>(p, l.flops, s);
  if (l.to == 64 && l.tc == 128) return launch_one<T, 64, 128, X3, ALIGNED, false>(p, l.flops, s);
  return launch_one<T, 64, 64, X3, ALIGNED, false>(p, l.flops, s);
}

int launch_wgrad(const WgradParams& p, const WgradLaunch& l, hipStream_t s) {
  if (p.K <= 0 || p.Cout <= 0 || p.Cin <= 0) return GC_OK;
  if (wgrad_image_applies(p, l)) return launch_wgrad_image(p, l, s);
  if (l.bf16) return l.aligned ? launch_tile<uint16_t, false, true>(p, l, s) : launch_tile<uint16_t, false, false>(p, l, s);
  if (l.x3) return l.aligned ? launch_tile<float, true, true>(p, l, s) : launch_tile<float, true, false>(p, l, s);
  return l.aligned ? launch_tile<float, false, true>(p, l, s) : launch_tile<float, false, false>(p, l, s);
}

int launch_wgrad_reduce(const WgradReduceParams& p0, hipStream_t stream) {
  WgradReduceParams p = p0;
  const size_t total = (size_t)p.Cout * (p.thin ? p.ICp : p.Cin) * p.kh * p.kw;
  if (total == 0) return GC_OK;
  const unsigned bias_blocks = p.bias_part != nullptr ? (unsigned)((p.Cout + 63) / 64) : 0u;
  const int T = p.kh * p.kw;
  if (!p.thin && T <= 16 && p.Cin % 4 == 0 && p.s_s == 1 && p.s_r == p.kw && p.s_ci == T && p.s_co == (int64_t)p.Cin * T) {
    const size_t pairs = (size_t)p.Cout * p.Cin;
    // enough waves to pull the slabs at full rate: more split waves per workgroup when there are few pairs
    const size_t wgs = (pairs + 63) / 64;
    p.main_blocks = (int32_t)wgs;
    const dim3 grid((unsigned)wgs + bias_blocks);
    if (p.splits >= 16 && wgs * 4 < 2048) hipLaunchKernelGGL(wgrad_reduce_dense_kernel<16>, grid, dim3(1024), 0, stream, p);
    else if (p.splits >= 4) hipLaunchKernelGGL(wgrad_reduce_dense_kernel<4>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(wgrad_reduce_dense_kernel<1>, grid, dim3(64), 0, stream, p);
    return check_hip(hipGetLastError(), "wgrad_reduce launch");
  }
  if (total % 4 == 0) {   // slabs stay 16-byte aligned: 4 elements per lane
    const size_t blocks = (total / 4 + 63) / 64;
    const unsigned threads = (p.splits >= 64 && blocks * 16 <= 4096) ? 1024u : (unsigned)kThreads;
    p.main_blocks = (int32_t)blocks;
    hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((unsigned)blocks + bias_blocks), dim3(threads), 0, stream, p);
  } else {
    const size_t blocks = (total + 63) / 64;
    const unsigned threads = (p.splits >= 64 && blocks * 16 <= 4096) ? 1024u : (unsigned)kThreads;
    p.main_blocks = (int32_t)blocks;
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)blocks + bias_blocks), dim3(threads), 0, stream, p);
  }
  return check_hip(hipGetLastError(), "wgrad_reduce launch");
}

int launch_colsum(const void* gy, float* gb, int K, int C, int cstride, int coff, bool bf16, hipStream_t stream) {
  if (zero_fill(gb, sizeof(float) * C, stream) != GC_OK) return GC_HIP_ERROR;
  if (K <= 0 || C <= 0) return GC_OK;
  int kpb = (K + 1023) / 1024;
  if (kpb < 64) kpb = 64;
  const int blocks = (K + kpb - 1) / kpb;
  if (bf16)
    hipLaunchKernelGGL(colsum_kernel<uint16_t>, dim3(blocks), dim3(kThreads), 0, stream, (const uint16_t*)gy, gb, K, C, cstride, coff, kpb);
  else
    hipLaunchKernelGGL(colsum_kernel<uint32_t>, dim3(blocks), dim3(kThreads), 0, stream, (const uint32_t*)gy, gb, K, C, cstride, coff, kpb);
  return check_hip(hipGetLastError(), "colsum launch");
}

}  // namespace gc
