// Self-attention over image positions, softmax(K Q^T) V, without materialising the [N, N] weights
// (reference BigGAN/model.py:97-103: bmm, softmax over the last dimension, bmm; N = 64 * 64 positions, key / query
// width C / 8, value width C).  bf16 operands, fp32 accumulation and softmax statistics.
//
//   K [B, N, d]  ("key": rows i of the weight matrix)        Q [B, N, d]  ("query": columns j)
//   V [B, N, C]  (rows j)                                     O [B, N, C] = softmax_j(K Q^T) V,  lse [B, N] = log sum_j exp
//
// Forward: a wave owns 32 rows i and walks the columns in blocks of 64.  It computes the TRANSPOSED scores
// S^T[j, i] = Q_j . K_i on the matrix cores, so that a lane's accumulators hold one column i (its softmax row) at
// j = 16 jb + 4 (lane / 16) + e.  Taking the 16x16x32 MFMA's reduction index in the order
// k = 8 g + e  <->  j = 32 q + 4 g + e,   k = 8 g + 4 + e  <->  j = 32 q + 16 + 4 g + e   (g = lane / 16)
// makes the exponentiated accumulators of blocks jb = 2q, 2q + 1 exactly the B operand P^T of O^T += V^T P^T: the
// probabilities go from accumulator to operand registers without a trip through LDS, and V^T in the same k order is
// what two transposing LDS reads (ds_read_b64_tr_b16: a 4 x 16 block, lane c gets column c) of the [j][c] tile return.
//
// Backward (dS = P (dP - delta), dP = dO V^T, delta_i = dO_i . O_i) recomputes P from lse in two kernels that each own
// the rows of their outputs, so there are no atomics and results are reproducible:
//   dqv: owner = 128 columns j, walks rows i:  dV_j = sum_i P_ij dO_i,   dQ_j = sum_i dS_ij K_i   (S with lanes = j)
//   dk : owner = 128 rows i, walks columns j:  dK_i = sum_j dS_ij Q_j                              (S^T with lanes = i)
// with the same accumulator -> operand trick for P and dS.
#include <stdio.h>

#include "common.h"

namespace gc {
namespace {

typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

struct AttnParams {
  const uint16_t *k, *q, *v, *o, *dout;
  uint16_t *out, *dk, *dq, *dv;
  float *lse, *delta;
  int B, N, d, C;
};

constexpr int kBlk = 64;      // positions staged per step
constexpr int kNarrow = 32;   // bytes per LDS row of a key / query tile: 16 bf16, zero-filled beyond d

__device__ __forceinline__ bf16x8 zero8() {
  const u32x4 z = {0u, 0u, 0u, 0u};
  return __builtin_bit_cast(bf16x8, z);
}
__device__ __forceinline__ bf16x8 pack8(f32x4 lo, f32x4 hi) {
  const u32x2 a = cvt_bf16x4(lo), b = cvt_bf16x4(hi);
  const u32x4 r = {a[0], a[1], b[0], b[1]};
  return __builtin_bit_cast(bf16x8, r);
}
// B-operand fragment of a row-major [rows][width] global tensor: 8 consecutive elements of row `row` from column 8 g
__device__ __forceinline__ bf16x8 row_frag(const uint16_t* base, size_t row, int width, int col) {
  if (col >= width) return zero8();
  return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(base + row * width + col));
}
// Transposed A-operand fragment from an LDS tile [row][col] (row pitch `pitch` bytes): for this lane's column
// col0 + lane % 16, the rows r0 + 4 g + {0..3} and r0 + 16 + 4 g + {0..3}.
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int r0, int col0_bytes, int lane) {
  const int row = r0 + 4 * (lane >> 4) + ((lane >> 2) & 3);
  const char* a0 = tile + row * pitch + col0_bytes + (lane & 3) * 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(a0 + 16 * pitch));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// exp(x) for a pair of scores already in units of log 2: two v_exp_f32
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kLog2e = 1.44269504088896340736f;
__device__ __forceinline__ f32x2 exp2_pair(f32x2 t) { return f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])}; }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
__device__ __forceinline__ bf16x8 ones8() {
  const u32x4 o = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  return __builtin_bit_cast(bf16x8, o);
}
// Exchanges across the wave's four 16-lane rows in the vector unit (v_permlane32_swap / v_permlane16_swap: with both
// operands the same register, one result holds the own half / row and the other the partner's), not through the LDS crossbar.
__device__ __forceinline__ float xor_max(float v) {
  typedef unsigned int u32pair __attribute__((ext_vector_type(2)));
  u32pair r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned int, v), __builtin_bit_cast(unsigned int, v), false, false);
  v = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
  r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned int, v), __builtin_bit_cast(unsigned int, v), false, false);
  return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

// Which (sample, tile of 128 positions) a workgroup owns.  Workgroup b runs on XCD b % 8, and every workgroup of a sample walks
// that sample's whole value / key tensors (0.5 - 1 MiB): the workgroups of one sample are given to ONE XCD, so that its L2
// (4 MiB) holds the two to four samples its CUs are working on; dealt round-robin, every L2 would see all 16 - 32 samples
// in flight and miss.  Samples beyond the last multiple of eight are dealt in order.
__device__ __forceinline__ void sample_and_tile(int B, int tiles, int& sample, int& tile) {
  const int b = blockIdx.x, grouped = (B & ~7) * tiles;
  if (b < grouped) {
    const int r = b >> 3;
    sample = (r / tiles) * 8 + (b & 7);
    tile = r % tiles;
  } else {
    const int r = b - grouped;
    sample = (B & ~7) + r / tiles;
    tile = r % tiles;
  }
}

// Stages 64 rows of a C-wide tensor and of a d-wide tensor (plus, optionally, 64 entries of two fp32 vectors) through
// registers into one LDS stage: wide tile rows of C * 2 + 32 bytes (the pad spreads the 16 rows a transposing read
// touches over the banks), narrow tile rows of 32 bytes.
template <int C, bool VEC>
struct Stager {
  static constexpr int WROW = C * 2 + 32;
  static constexpr int WBYTES = kBlk * WROW, NBYTES = kBlk * kNarrow;
  static constexpr int STAGE = WBYTES + NBYTES + (VEC ? 2 * kBlk * 4 : 0);
  static constexpr int CPR = C / 8;                      // 16-byte chunks per wide row
  static constexpr int WCH = kBlk * CPR / kThreads;      // chunks per thread
  u32x4 w[WCH], n;
  float f;
  __device__ __forceinline__ void fetch(const uint16_t* wide, const uint16_t* narrow, const float* va, const float* vb, size_t row0, int d,
                                        int tid) {
#pragma unroll
    for (int t = 0; t < WCH; t++) {
      const int ch = tid + kThreads * t;
      w[t] = *reinterpret_cast<const u32x4*>(wide + (row0 + ch / CPR) * C + (ch % CPR) * 8);
    }
    n = u32x4{0u, 0u, 0u, 0u};
    if (tid < 2 * kBlk && (tid & 1) * 8 < d) n = *reinterpret_cast<const u32x4*>(narrow + (row0 + (tid >> 1)) * d + (tid & 1) * 8);
    if (VEC) {
      f = 0.f;
      if (tid >= 128 && tid < 192) f = -va[row0 + tid - 128];   // negated: the consumers start their accumulators from them
      if (tid >= 192) f = -vb[row0 + tid - 192];
    }
  }
  __device__ __forceinline__ void stash(char* stage, int tid) const {
#pragma unroll
    for (int t = 0; t < WCH; t++) {
      const int ch = tid + kThreads * t;
      *reinterpret_cast<u32x4*>(stage + (ch / CPR) * WROW + (ch % CPR) * 16) = w[t];
    }
    if (tid < 2 * kBlk) *reinterpret_cast<u32x4*>(stage + WBYTES + (tid >> 1) * kNarrow + (tid & 1) * 16) = n;
    if (VEC && tid >= 128) *reinterpret_cast<float*>(stage + WBYTES + NBYTES + (tid - 128) * 4) = f;
  }
};

// ---------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(const AttnParams p) {
  typedef Stager<C, false> St;
  constexpr int NCB = C / 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * St::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  int sample, tile;
  sample_and_tile(p.B, p.N / 128, sample, tile);
  const size_t base = (size_t)sample * p.N;
  const int i0 = tile * 128 + wave * 32;
  const uint16_t* Q = p.q + base * p.d;
  const uint16_t* V = p.v + base * C;

  bf16x8 kfrag[2];
#pragma unroll
  for (int ib = 0; ib < 2; ib++) kfrag[ib] = row_frag(p.k + base * p.d, (size_t)(i0 + ib * 16 + c16), p.d, 8 * g);

  f32x4 acc[NCB][2];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int ib = 0; ib < 2; ib++) acc[cb][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[2] = {-INFINITY, -INFINITY};
  f32x4 accl[2] = {splat4(0.f), splat4(0.f)};        // sum_j p_ij, the same in every row of the block
  const bf16x8 ones = ones8();

  St st;
  st.fetch(V, Q, nullptr, nullptr, 0, p.d, tid);
  st.stash(smem, tid);
  __syncthreads();
  const int nblk = p.N / kBlk;
  for (int blk = 0; blk < nblk; blk++) {
    const bool more = blk + 1 < nblk;
    if (more) st.fetch(V, Q, nullptr, nullptr, (size_t)(blk + 1) * kBlk, p.d, tid);
    const char* vs = smem + (blk & 1) * St::STAGE;
    const char* qs = vs + St::WBYTES;

    f32x4 s[4][2];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qs + (jb * 16 + c16) * kNarrow + (g & 1) * 16);   // g >= 2: any finite data, the other operand is zero there
#pragma unroll
      for (int ib = 0; ib < 2; ib++)
        s[jb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kfrag[ib], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      float mx = m[ib];                    // three-operand maxima: two scores per instruction
#pragma unroll
      for (int jb = 0; jb < 4; jb++) {
        mx = fmaxf(fmaxf(s[jb][ib][0], s[jb][ib][1]), mx);
        mx = fmaxf(fmaxf(s[jb][ib][2], s[jb][ib][3]), mx);
      }
      const float mn = xor_max(mx);
      const float alpha = __builtin_amdgcn_exp2f((m[ib] - mn) * kLog2e);
      m[ib] = mn;
      // exp(s - mn) = exp2(s log2 e - mn log2 e): one multiply-add per score in front of the exponential; the row sums come
      // from the matrix cores (a block of ones beside V^T), which add the probabilities as the product uses them
      const f32x2 nb = {-mn * kLog2e, -mn * kLog2e}, sc = {kLog2e, kLog2e};
#pragma unroll
      for (int jb = 0; jb < 4; jb++) {
        const f32x2 lo = exp2_pair(f32x2{s[jb][ib][0], s[jb][ib][1]} * sc + nb);
        const f32x2 hi = exp2_pair(f32x2{s[jb][ib][2], s[jb][ib][3]} * sc + nb);
        s[jb][ib] = f32x4{lo[0], lo[1], hi[0], hi[1]};
      }
      if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {   // the running maximum rarely moves after the first blocks
        accl[ib] *= alpha;
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[cb][ib] *= alpha;
      }
    }
    bf16x8 pb[2][2];
#pragma unroll
    for (int qq = 0; qq < 2; qq++)
#pragma unroll
      for (int ib = 0; ib < 2; ib++) pb[qq][ib] = pack8(s[2 * qq][ib], s[2 * qq + 1][ib]);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
#pragma unroll
      for (int qq = 0; qq < 2; qq++) {
        const bf16x8 va = tr_frag(vs, St::WROW, 32 * qq, cb * 32, lane);
#pragma unroll
        for (int ib = 0; ib < 2; ib++) acc[cb][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[qq][ib], acc[cb][ib], 0, 0, 0);
      }
#pragma unroll
    for (int qq = 0; qq < 2; qq++)
#pragma unroll
      for (int ib = 0; ib < 2; ib++) accl[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[qq][ib], accl[ib], 0, 0, 0);
    if (more) st.stash(smem + ((blk + 1) & 1) * St::STAGE, tid);
    __syncthreads();
  }

#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    const float ls = accl[ib][0];
    const float inv = 1.f / ls;
    const size_t row = base + i0 + ib * 16 + c16;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++)
      *reinterpret_cast<u32x2*>(p.out + row * C + cb * 16 + g * 4) = cvt_bf16x4(acc[cb][ib] * inv);
    if (g == 0) p.lse[row] = m[ib] + __logf(ls);
  }
}

// delta[row] = sum_c dO[row, c] * O[row, c]; 16 lanes per row
__global__ __launch_bounds__(kThreads) void attn_delta_kernel(const uint16_t* o, const uint16_t* dout, float* delta, size_t rows, int C) {
  const size_t row = ((size_t)blockIdx.x * kThreads + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float s = 0.f;
  if (row < rows)
    for (int c = sub * 8; c < C; c += 128) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(o + row * C + c), b = *reinterpret_cast<const u32x4*>(dout + row * C + c);
      const f32x4 a0 = widen_bf16x4(u32x2{a[0], a[1]}), a1 = widen_bf16x4(u32x2{a[2], a[3]});
      const f32x4 b0 = widen_bf16x4(u32x2{b[0], b[1]}), b1 = widen_bf16x4(u32x2{b[2], b[3]});
#pragma unroll
      for (int e = 0; e < 4; e++) s += a0[e] * b0[e] + a1[e] * b1[e];
    }
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 8);
  if (row < rows && sub == 0) delta[row] = s;
}

// Owner: 128 columns j (32 per wave); walks the rows i in blocks of 64.  Produces dV and dQ.
template <int C>
__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dqv_kernel(const AttnParams p) {
  typedef Stager<C, true> St;
  constexpr int NCB = C / 16, NKS = C / 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * St::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  int sample, tile;
  sample_and_tile(p.B, p.N / 128, sample, tile);
  const size_t base = (size_t)sample * p.N;
  const int j0 = tile * 128 + wave * 32;
  const uint16_t* K = p.k + base * p.d;
  const uint16_t* dO = p.dout + base * C;
  const float* lse = p.lse + base;
  const float* delta = p.delta + base;

  bf16x8 qfrag[2], vfrag[2][NKS];
#pragma unroll
  for (int jb = 0; jb < 2; jb++) {
    const size_t j = (size_t)(j0 + jb * 16 + c16);
    qfrag[jb] = row_frag(p.q + base * p.d, j, p.d, 8 * g);
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) vfrag[jb][ks] = row_frag(p.v + base * C, j, C, ks * 32 + 8 * g);
  }
  f32x4 adv[NCB][2], adq[2];
#pragma unroll
  for (int jb = 0; jb < 2; jb++) {
    adq[jb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) adv[cb][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  St st;
  st.fetch(dO, K, lse, delta, 0, p.d, tid);
  st.stash(smem, tid);
  __syncthreads();
  const int nblk = p.N / kBlk;
  for (int blk = 0; blk < nblk; blk++) {
    const bool more = blk + 1 < nblk;
    if (more) st.fetch(dO, K, lse, delta, (size_t)(blk + 1) * kBlk, p.d, tid);
    const char* os = smem + (blk & 1) * St::STAGE;
    const char* ks_ = os + St::WBYTES;
    const float* ls = reinterpret_cast<const float*>(ks_ + St::NBYTES);

    f32x4 pm[4][2], ds[4][2];
#pragma unroll
    for (int ib = 0; ib < 4; ib++) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(ks_ + (ib * 16 + c16) * kNarrow + (g & 1) * 16);  // g >= 2: any finite data, the other operand is zero there
      bf16x8 oa[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) oa[ks] = *reinterpret_cast<const bf16x8*>(os + (ib * 16 + c16) * St::WROW + ks * 64 + g * 16);
      // -lse_i and -delta_i in the accumulator layout (row 4 g + e): the products start from them, so s - lse and
      // dP - delta come out of the matrix cores
      const f32x4 nli = *reinterpret_cast<const f32x4*>(ls + ib * 16 + 4 * g);
      const f32x4 ndi = *reinterpret_cast<const f32x4*>(ls + kBlk + ib * 16 + 4 * g);
      const f32x2 sc = {kLog2e, kLog2e};
#pragma unroll
      for (int jb = 0; jb < 2; jb++) {
        const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, qfrag[jb], nli, 0, 0, 0);
        f32x4 dp = ndi;
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(oa[ks], vfrag[jb][ks], dp, 0, 0, 0);
        const f32x2 p0 = exp2_pair(f32x2{s[0], s[1]} * sc), p1 = exp2_pair(f32x2{s[2], s[3]} * sc);
        const f32x2 d0 = p0 * f32x2{dp[0], dp[1]}, d1 = p1 * f32x2{dp[2], dp[3]};
        pm[ib][jb] = f32x4{p0[0], p0[1], p1[0], p1[1]};
        ds[ib][jb] = f32x4{d0[0], d0[1], d1[0], d1[1]};
      }
    }
#pragma unroll
    for (int qq = 0; qq < 2; qq++) {
      bf16x8 pp[2], pd[2];
#pragma unroll
      for (int jb = 0; jb < 2; jb++) {
        pp[jb] = pack8(pm[2 * qq][jb], pm[2 * qq + 1][jb]);
        pd[jb] = pack8(ds[2 * qq][jb], ds[2 * qq + 1][jb]);
      }
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
        const bf16x8 ot = tr_frag(os, St::WROW, 32 * qq, cb * 32, lane);
#pragma unroll
        for (int jb = 0; jb < 2; jb++) adv[cb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ot, pp[jb], adv[cb][jb], 0, 0, 0);
      }
      const bf16x8 kt = tr_frag(ks_, kNarrow, 32 * qq, 0, lane);
#pragma unroll
      for (int jb = 0; jb < 2; jb++) adq[jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, pd[jb], adq[jb], 0, 0, 0);
    }
    if (more) st.stash(smem + ((blk + 1) & 1) * St::STAGE, tid);
    __syncthreads();
  }

#pragma unroll
  for (int jb = 0; jb < 2; jb++) {
    const size_t row = base + j0 + jb * 16 + c16;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) *reinterpret_cast<u32x2*>(p.dv + row * C + cb * 16 + g * 4) = cvt_bf16x4(adv[cb][jb]);
    if (4 * g < p.d) *reinterpret_cast<u32x2*>(p.dq + row * p.d + g * 4) = cvt_bf16x4(adq[jb]);
  }
}

// Owner: 128 rows i (32 per wave); walks the columns j in blocks of 64.  Produces dK.
template <int C>
__global__ __launch_bounds__(kThreads, 2) void attn_bwd_dk_kernel(const AttnParams p) {
  typedef Stager<C, false> St;
  constexpr int NKS = C / 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * St::STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c16 = lane & 15;
  int sample, tile;
  sample_and_tile(p.B, p.N / 128, sample, tile);
  const size_t base = (size_t)sample * p.N;
  const int i0 = tile * 128 + wave * 32;
  const uint16_t* Q = p.q + base * p.d;
  const uint16_t* V = p.v + base * C;

  bf16x8 kfrag[2], ofrag[2][NKS];
  float li[2], di[2];
#pragma unroll
  for (int ib = 0; ib < 2; ib++) {
    const size_t i = (size_t)(i0 + ib * 16 + c16);
    kfrag[ib] = row_frag(p.k + base * p.d, i, p.d, 8 * g);
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) ofrag[ib][ks] = row_frag(p.dout + base * C, i, C, ks * 32 + 8 * g);
    li[ib] = -p.lse[base + i];           // negated: the products below start from them
    di[ib] = -p.delta[base + i];
  }
  f32x4 adk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  St st;
  st.fetch(V, Q, nullptr, nullptr, 0, p.d, tid);
  st.stash(smem, tid);
  __syncthreads();
  const int nblk = p.N / kBlk;
  for (int blk = 0; blk < nblk; blk++) {
    const bool more = blk + 1 < nblk;
    if (more) st.fetch(V, Q, nullptr, nullptr, (size_t)(blk + 1) * kBlk, p.d, tid);
    const char* vs = smem + (blk & 1) * St::STAGE;
    const char* qs = vs + St::WBYTES;

    f32x4 ds[4][2];
#pragma unroll
    for (int jb = 0; jb < 4; jb++) {
      const bf16x8 qa = *reinterpret_cast<const bf16x8*>(qs + (jb * 16 + c16) * kNarrow + (g & 1) * 16);   // g >= 2: any finite data, the other operand is zero there
      bf16x8 va[NKS];
#pragma unroll
      for (int ks = 0; ks < NKS; ks++) va[ks] = *reinterpret_cast<const bf16x8*>(vs + (jb * 16 + c16) * St::WROW + ks * 64 + g * 16);
#pragma unroll
      for (int ib = 0; ib < 2; ib++) {
        const f32x4 s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, kfrag[ib], splat4(li[ib]), 0, 0, 0);     // s - lse_i
        f32x4 dp = splat4(di[ib]);                                                                          // dP - delta_i
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[ks], ofrag[ib][ks], dp, 0, 0, 0);
        const f32x2 sc = {kLog2e, kLog2e};
        const f32x2 d0 = exp2_pair(f32x2{s[0], s[1]} * sc) * f32x2{dp[0], dp[1]}, d1 = exp2_pair(f32x2{s[2], s[3]} * sc) * f32x2{dp[2], dp[3]};
        ds[jb][ib] = f32x4{d0[0], d0[1], d1[0], d1[1]};
      }
    }
#pragma unroll
    for (int qq = 0; qq < 2; qq++) {
      const bf16x8 qt = tr_frag(qs, kNarrow, 32 * qq, 0, lane);
#pragma unroll
      for (int ib = 0; ib < 2; ib++)
        adk[ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, pack8(ds[2 * qq][ib], ds[2 * qq + 1][ib]), adk[ib], 0, 0, 0);
    }
    if (more) st.stash(smem + ((blk + 1) & 1) * St::STAGE, tid);
    __syncthreads();
  }
#pragma unroll
  for (int ib = 0; ib < 2; ib++)
    if (4 * g < p.d) *reinterpret_cast<u32x2*>(p.dk + (base + i0 + ib * 16 + c16) * p.d + g * 4) = cvt_bf16x4(adk[ib]);
}

int check_shape(int64_t B, int32_t N, int32_t d, int32_t C, const void* const* ptrs, int n) {
  if (B <= 0 || (int64_t)B * (N / 128) >= (1ll << 31) || N <= 0 || N % 128 != 0 || (d != 8 && d != 16) || (C != 64 && C != 128)) {
    set_error("attention: supported are N %% 128 == 0, d in {8, 16}, C in {64, 128}, B * N / 128 < 2^31 (got B=%lld N=%d d=%d C=%d)", (long long)B, N, d, C);
    return GC_UNSUPPORTED_GEOMETRY;
  }
  for (int i = 0; i < n; i++)
    if (ptrs[i] == nullptr || (reinterpret_cast<uintptr_t>(ptrs[i]) & 15) != 0) {
      set_error("attention: every tensor must be non-null and 16-byte aligned");
      return GC_BAD_ARGUMENT;
    }
  return GC_OK;
}

}  // namespace
}  // namespace gc

using namespace gc;

extern "C" int gc_attention_fwd(const void* k, const void* q, const void* v, void* out, float* lse, int32_t B, int32_t N, int32_t d, int32_t C,
                                void* stream) {
  const void* ptrs[] = {k, q, v, out, lse};
  if (int e = check_shape(B, N, d, C, ptrs, 5)) return e;
  AttnParams p{};
  p.k = (const uint16_t*)k; p.q = (const uint16_t*)q; p.v = (const uint16_t*)v; p.out = (uint16_t*)out; p.lse = lse;
  p.B = B; p.N = N; p.d = d; p.C = C;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps("gc::attn_fwd_kernel", 2.0 * B * N * (double)N * (d + C), s);
  const dim3 grid((unsigned)(N / 128) * (unsigned)B);
  if (C == 128) hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(kThreads), 0, s, p);
  else hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(kThreads), 0, s, p);
  return check_hip(hipGetLastError(), "attn_fwd launch");
}

extern "C" int gc_attention_bwd(const void* k, const void* q, const void* v, const void* out, const void* dout, const float* lse, void* dk,
                                void* dq, void* dv, float* delta, int32_t B, int32_t N, int32_t d, int32_t C, void* stream) {
  const void* ptrs[] = {k, q, v, out, dout, lse, dk, dq, dv, delta};
  if (int e = check_shape(B, N, d, C, ptrs, 10)) return e;
  AttnParams p{};
  p.k = (const uint16_t*)k; p.q = (const uint16_t*)q; p.v = (const uint16_t*)v; p.o = (const uint16_t*)out; p.dout = (const uint16_t*)dout;
  p.dk = (uint16_t*)dk; p.dq = (uint16_t*)dq; p.dv = (uint16_t*)dv; p.lse = const_cast<float*>(lse); p.delta = delta;
  p.B = B; p.N = N; p.d = d; p.C = C;
  hipStream_t s = (hipStream_t)stream;
  const size_t rows = (size_t)B * N;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((rows * 16 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, p.o, p.dout, delta, rows, C);
  if (check_hip(hipGetLastError(), "attn_delta launch")) return GC_HIP_ERROR;
  const dim3 grid((unsigned)(N / 128) * (unsigned)B);
  {
    ProfScope ps("gc::attn_bwd_dqv_kernel", 2.0 * B * N * (double)N * (2 * d + 2 * C), s);
    if (C == 128) hipLaunchKernelGGL(attn_bwd_dqv_kernel<128>, grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL(attn_bwd_dqv_kernel<64>, grid, dim3(kThreads), 0, s, p);
    if (check_hip(hipGetLastError(), "attn_bwd_dqv launch")) return GC_HIP_ERROR;
  }
  {
    ProfScope ps("gc::attn_bwd_dk_kernel", 2.0 * B * N * (double)N * (2 * d + C), s);
    if (C == 128) hipLaunchKernelGGL(attn_bwd_dk_kernel<128>, grid, dim3(kThreads), 0, s, p);
    else hipLaunchKernelGGL(attn_bwd_dk_kernel<64>, grid, dim3(kThreads), 0, s, p);
  }
  return check_hip(hipGetLastError(), "attn_bwd_dk launch");
}
