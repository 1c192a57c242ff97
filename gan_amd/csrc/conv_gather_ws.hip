// Persistent, wave-specialised gather-GEMM convolution (forward and data gradient) for bf16 activations: the kernel of
// the mid-size layers (17-70 GFLOP), where a workgroup's fixed costs -- reading its arguments, decomposing pixel
// indices, the first loads' latency, the output stores -- are as long as its multiply phase.
//
// Same tile, LDS image and LDS-DMA addressing as conv_gather_dma.hip (TO channels x 128 pixels, 64-deep steps, 128-byte
// rows XOR-swizzled on the source side).  What differs:
//   * A workgroup is 4 consumer waves (fragment reads, MFMA, epilogue) + 4 loader waves (LDS-DMA only) and walks
//     SEVERAL tiles (tile = blockIdx.x, += gridDim.x, XCD-contiguous inside a round).  The ring of STAGES buffers runs
//     across tile boundaries: while the consumers store tile i, the loaders already have the first steps of tile i + 1
//     in flight, and their own per-tile set-up (pixel decomposition of their rows) hides behind the consumers' loop.
//   * Arguments are packed for the scalar unit: one 32-byte record per output class fetched with one scalar load, taps
//     as one dword each, divisions by multiply-shift.  A wave alone on its SIMD issues a vector instruction every 4-5
//     cycles and every dependent scalar load costs a round trip to memory, so the per-tile code is kept to a few
//     hundred instructions.
//   * The epilogue is the branch-free path only (full channel tiles, 16-byte aligned rows, none / ReLU / LeakyReLU);
//     everything else stays on conv_gather_dma.hip.  Weight rows are permuted inside groups of 32 (oc_of_row) so that a
//     lane holds 8 consecutive channels of a pixel and stores them with one 16-byte instruction.
//
// Synchronisation: barrier #s (all 8 waves) separates "step s has landed" (the loaders waited for it with a counted
// vmcnt before arriving) from "step s - 1 has been consumed" (the consumers arrive after their MFMAs of step s - 1
// were issued, i.e. after its fragment reads returned); behind it the loaders refill the buffer of step s - 1 with step
// s + STAGES - 1.  Both roles derive the number of steps of a tile from the same class record, so they execute the same
// number of barriers.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "conv_gather.h"

namespace gc {
namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
constexpr uint32_t kOutOfRange = 0x80000000u;
constexpr int kTP = 128;      // pixels per tile
constexpr int kNW = 4;        // waves per role

struct WsClass {              // 32 bytes: one scalar load
  int32_t tile_start;         // first pixel tile of the class
  int32_t Mc;                 // pixels of the class: N * oh * ow
  uint32_t oh_ow;             // oh | ow << 16
  uint32_t ph_pw_taps;        // ph | pw << 8 | tap_begin << 16 | tap_end << 24
  uint32_t ow_mul, ow_shift;  // division by ow: mul == 0 -> shift only (power of two)
  uint32_t oh_mul, oh_shift;
};

struct WsParams {
  const void* in;
  const uint16_t* wp;
  void* out;
  const float* bias;
  float* ch_sum;
  float* ch_sumsq;
  float* stat_rows;                 // null, or [pixel tile * WP + wave's pixel part][OC][2]: (sum, sum of squares) of the stored values
  uint16_t* out_bf16;
  const void* mask;                 // null, or indexed like `out`: the stored value is scaled by (mask > 0 ? 1 : mask_slope)
  unsigned long long* diag_buf;
  int32_t in_bytes, w_bytes, a_bias;
  int32_t IH, IW, IC, csteps, OH, OW, row_pitch, ICp;
  int32_t a, b;
  int32_t in_cstride, in_coff, out_cstride, out_coff;
  int32_t act;
  float slope, mask_slope;
  int32_t n_tiles, n_oc_tiles, n_classes, cls_interleave;
  // split reduction (SPLIT variants): a tile's steps are dealt to `ksplit` workgroups, each writing its accumulators to
  // `part` in fragment order; a second launch of the same kernel (`nsum` = the number of splits, ksplit = 1) adds them
  // in a fixed order and runs the ordinary epilogue
  int32_t ksplit, steps_per_split, nsum;
  float* part;
  uint32_t ks_mul, ks_shift;        // division by ksplit
  uint32_t oct_mul, oct_shift;      // division by n_oc_tiles
  uint32_t ncls_mul, ncls_shift;    // division by n_classes
  WsClass cls[kMaxClasses];
  int32_t taps[kMaxTaps];           // (dh & 0xff) | (dw & 0xff) << 8 | wtap << 16
};

__device__ __forceinline__ int dswz(int row) { return (row >> 1) & 7; }

// channel (relative to the tile) held by the weight row at LDS row `row`: see the file header
__device__ __forceinline__ int oc_of_row(int row) {
  const int f = (row >> 4) & 1, g = (row >> 2) & 3, e = row & 3;
  return (row & ~31) + 8 * g + 4 * f + e;
}

// n / d for n < 2^31 with (mul, shift) from ws_div(): mul == 0 marks a power of two
__device__ __forceinline__ uint32_t wdiv(uint32_t n, uint32_t mul, uint32_t shift) {
  const uint32_t q = __umulhi(n, mul);
  const uint32_t t = ((n - q) >> 1) + q;
  return (mul == 0 ? n : t) >> shift;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

struct TileId {
  int cls, m0, oc0, tile, split;
};

template <bool SPLIT>
__device__ __forceinline__ bool tile_of(const WsParams& p, int k, TileId& id) {
  const int G = (int)gridDim.x;
  const int base = k * G;
  const int units = SPLIT ? p.n_tiles * p.ksplit : p.n_tiles;
  const int nr = min(G, units - base);            // tiles (or tile parts) of this round
  if ((int)blockIdx.x >= nr) return false;
  uint32_t lid = (uint32_t)base + xcd_remap(blockIdx.x, (uint32_t)nr);
  id.split = 0;
  if (SPLIT) {                 // the parts of one tile are neighbours: they read the same pixels and weights
    const uint32_t q = wdiv(lid, p.ks_mul, p.ks_shift);
    id.split = (int)(lid - q * (uint32_t)p.ksplit);
    lid = q;
  }
  id.tile = (int)lid;
  const uint32_t pix_tile = wdiv(lid, p.oct_mul, p.oct_shift);
  id.oc0 = (int)(lid - pix_tile * (uint32_t)p.n_oc_tiles);
  if (p.cls_interleave) {       // equal-sized classes in (tile, class) order: the classes of one image region are neighbours
    const uint32_t q = wdiv(pix_tile, p.ncls_mul, p.ncls_shift);
    id.cls = (int)(pix_tile - q * (uint32_t)p.n_classes);
    id.m0 = (int)q * kTP;
  } else {
    int cls = 0;
    for (int c = 1; c < p.n_classes; c++)
      if ((int)pix_tile >= p.cls[c].tile_start) cls = c;
    id.cls = cls;
    id.m0 = ((int)pix_tile - p.cls[cls].tile_start) * kTP;
  }
  return true;
}

// first step and number of steps of a tile (or of one part of it)
template <bool SPLIT>
__device__ __forceinline__ int steps_of(const WsParams& p, uint32_t ph_pw_taps, int split, int& s0) {
  const int total = (int)((ph_pw_taps >> 24) - ((ph_pw_taps >> 16) & 0xff)) * p.csteps;
  s0 = 0;
  if (!SPLIT) return total;
  if (p.nsum > 0) return 0;             // the adding launch has no steps
  s0 = split * p.steps_per_split;
  return min(p.steps_per_split, total - s0);
}

// MASK: the epilogue scales the stored value by the derivative selected by p.mask (a variant of its own: the extra loads
// cost the 128-channel tile the register budget of its second workgroup per CU)
template <typename TOut, int TO, int STAGES, bool MASK = false, bool SPLIT = false>
__global__ __launch_bounds__(2 * kNW * 64) void conv_gather_ws_kernel(const WsParams p) {
  constexpr int WO = TO >= 128 ? 2 : 1, WP = kNW / WO;
  constexpr int FO = TO / WO / 16, FP = kTP / WP / 16;
  constexpr int W_BYTES = TO * 128, A_BYTES = kTP * 128, STAGE = W_BYTES + A_BYTES;
  constexpr int AJ = kTP / 8 / kNW, WJ = TO / 8 / kNW, LOADS = AJ + WJ;
  static_assert(FO % 2 == 0, "the epilogue stores pairs of 16-row blocks");
  static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 2) * LOADS < 64, "ring depth / vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* tap_lds = reinterpret_cast<int*>(smem + STAGES * STAGE);

  const int t = threadIdx.x;
  const int lane = t & 63, wave_id = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave_id >= kNW;
  const int wave = loader ? wave_id - kNW : wave_id;
  const unsigned long long stamp0 = p.diag_buf != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;

  if (t < kMaxTaps) tap_lds[t] = p.taps[t];
  __syncthreads();

  auto next_stage = [](int s) { return s + 1 == STAGES ? 0 : s + 1; };

  if (loader) {
    // ================================================================ loader waves
    if (SPLIT && p.nsum > 0) return;      // the adding launch loads nothing (and has no barriers: its tiles have no steps)
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) - p.a_bias, 0, p.in_bytes + p.a_bias, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.wp), 0, p.w_bytes, 0x00020000);
    const int sub_row = lane >> 3, pos = lane & 7;

    // ---- issue cursor: (tile k_i, tap, channel step) of the next step to load, with the tile's per-lane rows
    uint32_t a_off[AJ], a_voff[AJ], w_voff[WJ];     // a_voff: a_off, or out of range where the current tap leaves the image
    const bool full_steps = p.IC % kBK == 0;       // every 64-channel step lies inside the tensor's channels
    int a_ih[AJ], a_iw[AJ], a_chunk[AJ];
    int k_i = 0, tap_i = 0, cs_i = 0, left_i = 0;
    int s_dh = 0, s_dw = 0;
    uint32_t s_aoff = 0, s_woff = 0;
    bool have = false;
    auto load_tap = [&](int tp) {
      const int e = __builtin_amdgcn_readfirstlane(tap_lds[tp]);
      s_dh = (int)(int8_t)(e & 0xff);
      s_dw = (int)(int8_t)((e >> 8) & 0xff);
      s_aoff = (uint32_t)(((s_dh * p.IW + s_dw) * p.in_cstride) * 2 + p.a_bias);
      s_woff = (uint32_t)((e >> 16) * p.ICp * 2);
      // which rows the tap keeps inside the image: once per tap, not once per channel step
#pragma unroll
      for (int j = 0; j < AJ; j++) {
        // (bitwise on purpose: a short-circuit would turn into exec-mask branches)
        const bool ok = ((unsigned)(a_ih[j] + s_dh) < (unsigned)p.IH) & ((unsigned)(a_iw[j] + s_dw) < (unsigned)p.IW);
        a_voff[j] = ok ? a_off[j] : kOutOfRange;
      }
    };
    auto setup_tile = [&](int k) {
      TileId id;
      have = tile_of<SPLIT>(p, k, id);
      if (!have) return;
      const WsClass c = p.cls[id.cls];
      const int OHc = (int)(c.oh_ow & 0xffff), OWc = (int)(c.oh_ow >> 16);
      int s0;
      left_i = steps_of<SPLIT>(p, c.ph_pw_taps, id.split, s0);
      const int t0 = SPLIT ? s0 / p.csteps : 0;
      tap_i = (int)((c.ph_pw_taps >> 16) & 0xff) + t0;
      cs_i = SPLIT ? s0 - t0 * p.csteps : 0;
#pragma unroll
      for (int j = 0; j < AJ; j++) {
        const int row = (wave * AJ + j) * 8 + sub_row;
        const int m = id.m0 + row;
        const bool ok = m < c.Mc;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        const uint32_t tmp = wdiv(mm, c.ow_mul, c.ow_shift), n = wdiv(tmp, c.oh_mul, c.oh_shift);
        const int ow = (int)(mm - tmp * (uint32_t)OWc), oh = (int)(tmp - n * (uint32_t)OHc);
        a_ih[j] = ok ? oh * p.a : -(1 << 20);
        a_iw[j] = ow * p.a;
        a_chunk[j] = (pos ^ dswz(row)) * 8;
        a_off[j] = (uint32_t)((((size_t)n * p.IH + oh * p.a) * p.IW + ow * p.a) * p.in_cstride + p.in_coff + a_chunk[j]) * 2u;
      }
#pragma unroll
      for (int j = 0; j < WJ; j++) {
        const int row = (wave * WJ + j) * 8 + sub_row;
        const int oc = id.oc0 * TO + oc_of_row(row);
        w_voff[j] = (uint32_t)(((size_t)oc * p.row_pitch + (pos ^ dswz(row)) * 8) * 2);
      }
      load_tap(tap_i);
    };
    int st_i = 0, issued = 0;
    auto advance = [&]() {
      issued++;
      if (--left_i == 0) {
        setup_tile(++k_i);
      } else if (++cs_i == p.csteps) {
        cs_i = 0;
        load_tap(++tap_i);
      }
    };
    setup_tile(0);

    {
      auto issue_next = [&]() {
        if (!have) return;
        char* wb = smem + st_i * STAGE;
        char* ab = wb + W_BYTES;
        const int c0 = cs_i * kBK;
#pragma unroll
        for (int j = 0; j < WJ; j++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(wb + (wave * WJ + j) * 1024), 16, (int)w_voff[j],
                                                   (int)(s_woff + c0 * 2), 0, 0);
#pragma unroll
        for (int j = 0; j < AJ; j++) {
          const uint32_t voff = (full_steps || (c0 | a_chunk[j]) < p.IC) ? a_voff[j] : kOutOfRange;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(ab + (wave * AJ + j) * 1024), 16, (int)voff,
                                                   (int)(s_aoff + c0 * 2), 0, 0);
        }
        st_i = next_stage(st_i);
        advance();
      };
#pragma unroll
      for (int s = 0; s < STAGES - 1; s++) issue_next();
      // ---- barrier cursor: every step of every tile of this workgroup, in order
      int done = 0;
      unsigned long long t_vm = 0, t_bar = 0, t_iss = 0;
      const bool dg = p.diag_buf != nullptr;
      for (int k = 0;; k++) {
        TileId id;
        if (!tile_of<SPLIT>(p, k, id)) break;
        int s0;
        const int nsteps = steps_of<SPLIT>(p, p.cls[id.cls].ph_pw_taps, id.split, s0);
        for (int it = 0; it < nsteps; it++, done++) {
          const int younger = issued - done - 1;      // steps issued behind step `done`: they may stay in flight
          const unsigned long long q0 = dg ? __builtin_amdgcn_s_memtime() : 0ull;
          if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * LOADS>();
          else if (STAGES >= 3 && younger >= 1) wait_vmcnt<LOADS>();
          else wait_vmcnt<0>();
          const unsigned long long q1 = dg ? __builtin_amdgcn_s_memtime() : 0ull;
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long q2 = dg ? __builtin_amdgcn_s_memtime() : 0ull;
          issue_next();
          if (dg) { t_vm += q1 - q0; t_bar += q2 - q1; t_iss += __builtin_amdgcn_s_memtime() - q2; }
        }
      }
      if (dg && wave == 0 && lane == 0) {
        unsigned long long* d = p.diag_buf + (size_t)(gridDim.x + blockIdx.x) * 4;
        d[0] = t_vm; d[1] = t_bar; d[2] = t_iss; d[3] = 0;
      }
    }
    return;
  }

  // ================================================================== consumer waves
  const int wo = wave / WP, wp_ = wave % WP;
  const int g = lane >> 4, c16 = lane & 15;
  TOut* out = reinterpret_cast<TOut*>(p.out);
  const TOut* mask = reinterpret_cast<const TOut*>(p.mask);
  const bool stats = !MASK && (p.ch_sum != nullptr || p.stat_rows != nullptr);      // (a masked gradient never carries statistics)
  const float sl = p.act == GC_ACT_NONE ? 1.f : (p.act == GC_ACT_RELU ? 0.f : p.slope);   // v > 0 ? v : v * sl
  unsigned long long epi_cycles = 0, wait_cycles = 0;
  int st_c = 0;

  for (int k = 0;; k++) {
    TileId id;
    if (!tile_of<SPLIT>(p, k, id)) break;
    const WsClass c = p.cls[id.cls];
    int s0_unused;
    const int nsteps = steps_of<SPLIT>(p, c.ph_pw_taps, id.split, s0_unused);

    f32x4 acc[FO][FP];
#pragma unroll
    for (int i = 0; i < FO; i++)
#pragma unroll
      for (int j = 0; j < FP; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int it = 0; it < nsteps; it++) {
      const unsigned long long ta = p.diag_buf != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (p.diag_buf != nullptr) wait_cycles += __builtin_amdgcn_s_memtime() - ta;
      const char* wb = smem + st_c * STAGE;
      const char* ab = wb + W_BYTES;
      // Fragment reads are issued in batches ahead of the MFMAs that use them (the scheduler would otherwise sink every
      // read next to its first use and expose one LDS latency per pair of reads): a whole step for the small tile,
      // one 32-deep half step for the large one.  The co-resident workgroup's consumer wave on the same SIMD fills the
      // matrix pipe while this wave waits for a batch.
      bf16x8 a[2][FO], b[2][FP];
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int kc = ks * 4 + g;
#pragma unroll
        for (int f = 0; f < FO; f++) {
          const int row = wo * (TO / WO) + f * 16 + c16;
          a[ks][f] = *reinterpret_cast<const bf16x8*>(wb + row * 128 + ((kc ^ dswz(row)) << 4));
        }
#pragma unroll
        for (int f = 0; f < FP; f++) {
          const int row = wp_ * (kTP / WP) + f * 16 + c16;
          b[ks][f] = *reinterpret_cast<const bf16x8*>(ab + row * 128 + ((kc ^ dswz(row)) << 4));
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int i = 0; i < FO; i++)
#pragma unroll
          for (int j = 0; j < FP; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks][i], b[ks][j], acc[i][j], 0, 0, 0);
      if constexpr (FO * FP <= 8) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (FO + FP), 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * FO * FP, 0);
      } else {
        __builtin_amdgcn_sched_group_barrier(0x100, FO + FP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FO * FP, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, FO + FP, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, FO * FP, 0);
      }
      st_c = next_stage(st_c);
      __builtin_amdgcn_sched_barrier(0);
    }

    if (SPLIT) {
      // fragment-order table: [tile][part][fragment][consumer thread] x 4 floats, i.e. 1 KB per store instruction
      constexpr int NF = FO * FP;
      if (p.nsum == 0) {          // a part of the reduction: the accumulators as they are, no epilogue
        float* dst = p.part + (((size_t)id.tile * p.ksplit + id.split) * NF * (kNW * 64) + (size_t)t) * 4;
#pragma unroll
        for (int i = 0; i < FO; i++)
#pragma unroll
          for (int j = 0; j < FP; j++) *reinterpret_cast<f32x4*>(dst + (size_t)(i * FP + j) * (kNW * 64) * 4) = acc[i][j];
        continue;
      }
      for (int sp = 0; sp < p.nsum; sp++) {       // the adding launch: the parts in their order, then the ordinary epilogue
        const float* src = p.part + (((size_t)id.tile * p.nsum + sp) * NF * (kNW * 64) + (size_t)t) * 4;
#pragma unroll
        for (int i = 0; i < FO; i++)
#pragma unroll
          for (int j = 0; j < FP; j++) acc[i][j] += *reinterpret_cast<const f32x4*>(src + (size_t)(i * FP + j) * (kNW * 64) * 4);
      }
    }

    // ---- epilogue of this tile
    const unsigned long long e0 = p.diag_buf != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
    const int OHc = (int)(c.oh_ow & 0xffff), OWc = (int)(c.oh_ow >> 16);
    const int ph = (int)(c.ph_pw_taps & 0xff), pw = (int)((c.ph_pw_taps >> 8) & 0xff);
    const int oc_w = id.oc0 * TO + wo * (TO / WO);
    const bool linear = p.n_classes == 1 && p.b == 1;       // forward pass: output pixel index == GEMM column index
    f32x4 bias_v[FO];
#pragma unroll
    for (int i = 0; i < FO; i++) {
      const int oc = oc_w + (i >> 1) * 32 + g * 8 + (i & 1) * 4;
      bias_v[i] = p.bias != nullptr ? *reinterpret_cast<const f32x4*>(p.bias + oc) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float ssum[FO][4], ssq[FO][4];
#pragma unroll
    for (int i = 0; i < FO; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) ssum[i][r] = ssq[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < FP; j++) {
      const uint32_t m = (uint32_t)(id.m0 + wp_ * (kTP / WP) + j * 16 + c16);
      const bool mok = (int)m < c.Mc;
      size_t pix = m;
      if (!linear) {
        const uint32_t mm = mok ? m : 0u;
        const uint32_t tmp = wdiv(mm, c.ow_mul, c.ow_shift), n = wdiv(tmp, c.oh_mul, c.oh_shift);
        const int ow = (int)(mm - tmp * (uint32_t)OWc), oh = (int)(tmp - n * (uint32_t)OHc);
        pix = ((size_t)n * p.OH + (oh * p.b + ph)) * p.OW + (ow * p.b + pw);
      }
      const size_t eoff = pix * p.out_cstride + p.out_coff + oc_w + g * 8;
#pragma unroll
      for (int i = 0; i < FO; i += 2) {
        f32x4 va = acc[i][j] + bias_v[i], vb = acc[i + 1][j] + bias_v[i + 1];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          va[r] = va[r] > 0.f ? va[r] : va[r] * sl;
          vb[r] = vb[r] > 0.f ? vb[r] : vb[r] * sl;
        }
        if (MASK) {                     // derivative of the activation whose output is `mask`
          f32x4 ma, mb;
          const size_t moff = mok ? eoff + (i >> 1) * 32 : 0;
          if (sizeof(TOut) == 2) {
            const u32x4 mv = *reinterpret_cast<const u32x4*>(mask + moff);
            ma = widen_bf16x4(u32x2{mv[0], mv[1]});
            mb = widen_bf16x4(u32x2{mv[2], mv[3]});
          } else {
            ma = *reinterpret_cast<const f32x4*>(mask + moff);
            mb = *reinterpret_cast<const f32x4*>(mask + moff + 4);
          }
#pragma unroll
          for (int r = 0; r < 4; r++) {
            va[r] = ma[r] > 0.f ? va[r] : va[r] * p.mask_slope;
            vb[r] = mb[r] > 0.f ? vb[r] : vb[r] * p.mask_slope;
          }
        }
        const u32x2 pa = cvt_bf16x4(va), pb = cvt_bf16x4(vb);
        if (mok) {
          if (sizeof(TOut) == 2) {
            *reinterpret_cast<u32x4*>(out + eoff + (i >> 1) * 32) = u32x4{pa[0], pa[1], pb[0], pb[1]};
          } else {
            *reinterpret_cast<f32x4*>(out + eoff + (i >> 1) * 32) = va;
            *reinterpret_cast<f32x4*>(out + eoff + (i >> 1) * 32 + 4) = vb;
            if (p.out_bf16 != nullptr) *reinterpret_cast<u32x4*>(p.out_bf16 + eoff + (i >> 1) * 32) = u32x4{pa[0], pa[1], pb[0], pb[1]};
          }
        }
        if (stats && mok) {
          if (sizeof(TOut) == 2) { va = widen_bf16x4(pa); vb = widen_bf16x4(pb); }     // statistics of the stored values
#pragma unroll
          for (int r = 0; r < 4; r++) {
            ssum[i][r] += va[r]; ssq[i][r] += va[r] * va[r];
            ssum[i + 1][r] += vb[r]; ssq[i + 1][r] += vb[r] * vb[r];
          }
        }
      }
    }
    if (stats) {
#pragma unroll
      for (int i = 0; i < FO; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float s = ssum[i][r], q = ssq[i][r];
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) {
            s += __shfl_xor(s, d, 64);
            q += __shfl_xor(q, d, 64);
          }
          const int oc = oc_w + (i >> 1) * 32 + g * 8 + (i & 1) * 4 + r;
          if (c16 == 0) {
            if (p.stat_rows != nullptr) {      // this wave's pixels of this tile are one row of the partial table
              const size_t row = (size_t)((id.m0 / kTP + p.cls[id.cls].tile_start) * WP + wp_);
              *reinterpret_cast<float2*>(p.stat_rows + (row * (size_t)(p.n_oc_tiles * TO) + oc) * 2) = float2{s, q};
            } else {
              atomicAdd(p.ch_sum + oc, s);
              if (p.ch_sumsq != nullptr) atomicAdd(p.ch_sumsq + oc, q);
            }
          }
        }
    }
    if (p.diag_buf != nullptr) epi_cycles += __builtin_amdgcn_s_memtime() - e0;
  }
  if (p.diag_buf != nullptr && t == 0) {
    unsigned long long* d = p.diag_buf + (size_t)blockIdx.x * 4;
    d[0] = stamp0; d[1] = __builtin_amdgcn_s_memtime(); d[2] = epi_cycles; d[3] = wait_cycles;
  }
}

// (mul, shift) for wdiv: exact for n < 2^31
void ws_div(uint32_t d, uint32_t* mul, uint32_t* shift) {
  const FastDiv f = make_fastdiv(d);
  *mul = f.pow2 ? 0u : f.mul;
  *shift = f.shift;
}

template <typename TOut, int TO, int STAGES, bool MASK = false, bool SPLIT = false>
int launch_one(const WsParams& p, int grid, double flops, hipStream_t stream) {
  constexpr int smem = STAGES * (TO + kTP) * 128 + kMaxTaps * 4;
  auto kern = conv_gather_ws_kernel<TOut, TO, STAGES, MASK, SPLIT>;
  if (smem > 64 * 1024 && ensure_dynamic_smem(reinterpret_cast<const void*>(kern), smem) != GC_OK) return GC_HIP_ERROR;
  char name[128];
  // (the demangled kernel name as a profiler prints it, plus which of its two launches a split reduction is)
  snprintf(name, sizeof(name), "gc::conv_gather_ws_kernel<%s, %d, %d, %s, %s>%s", sizeof(TOut) == 4 ? "float" : "unsigned short", TO, STAGES,
           MASK ? "true" : "false", SPLIT ? "true" : "false", !SPLIT ? "" : (p.nsum > 0 ? " (adding the parts)" : " (split reduction)"));
#ifdef GC_DIAG    // diagnostic build only (make DIAG=1): this block allocates and synchronises, which the shipped library never does
  static const bool stamps = getenv("GC_DIAG_STAMPS") != nullptr;
  if (stamps) {   // synchronous launches with cycle stamps per workgroup, summary on stderr
    static unsigned long long* dbuf = nullptr;
    static unsigned long long hbuf[2 * 2048 * 4];
    if (dbuf == nullptr) (void)hipMalloc(&dbuf, sizeof(hbuf));
    static int count = 0;
    if (grid <= 2048 && dbuf != nullptr && ++count > 3 && count <= 6) {
      WsParams q = p;
      q.diag_buf = dbuf;
      hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * kNW * 64), smem, stream, q);
      (void)hipStreamSynchronize(stream);
      (void)hipMemcpy(hbuf, dbuf, (size_t)grid * 64, hipMemcpyDeviceToHost);
      double life = 0, epi = 0, wait = 0, lvm = 0, lbar = 0, liss = 0;
      for (int b = 0; b < grid; b++) {
        life += (double)(hbuf[b * 4 + 1] - hbuf[b * 4]);
        epi += (double)hbuf[b * 4 + 2];
        wait += (double)hbuf[b * 4 + 3];
        lvm += (double)hbuf[(grid + b) * 4]; lbar += (double)hbuf[(grid + b) * 4 + 1]; liss += (double)hbuf[(grid + b) * 4 + 2];
      }
      fprintf(stderr, "[stamps] loader wave 0: waiting for loads %.0f, at barriers %.0f, issuing %.0f cycles\n", lvm / grid, lbar / grid, liss / grid);
      fprintf(stderr, "[stamps] %s grid %d tiles %d: lifetime %.0f cycles per workgroup, of which epilogues %.0f (%.0f per tile), consumer wave 0 at barriers %.0f\n",
              name, grid, p.n_tiles, life / grid, epi / grid, epi / p.n_tiles, wait / grid);
      return GC_OK;
    }
  }
#endif
  ProfScope ps(name, flops, stream);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(2 * kNW * 64), smem, stream, p);
  return check_hip(hipGetLastError(), "conv_gather_ws launch");
}

}  // namespace

// Takes the problems the branch-free epilogue covers (see the file header); false leaves p and l untouched.
bool plan_ws(const GatherParams& p, const GatherLaunch& l) {
  static const bool off = getenv("GC_NO_WS") != nullptr;
  if (off) return false;
  if (!l.in_bf16 || l.mode != kVecLoads || p.ksplit != 1 || l.tp != kTP) return false;
  if (l.to != 64 && l.to != 128) return false;
  if (p.OC % l.to != 0 || !p.out_vec8 || p.act > GC_ACT_LRELU) return false;
  if (p.tt.n_classes > kMaxClasses || p.tt.cls_tap_start[p.tt.n_classes] > kMaxTaps) return false;
  return true;
}

int launch_ws(const GatherParams& g, const GatherLaunch& l, bool out_f32, hipStream_t stream) {
  if (g.n_pix_tiles == 0 || g.n_oc_tiles == 0) return GC_OK;
  WsParams p;
  memset(&p, 0, sizeof(p));
  p.in = g.in; p.wp = g.wp; p.out = g.out; p.bias = g.bias; p.ch_sum = g.ch_sum; p.ch_sumsq = g.ch_sumsq; p.out_bf16 = g.out_bf16;
  if (g.stat_rows != nullptr) {      // one row per consumer wave's pixel share of a tile (4 or 2 per 128-pixel tile)
    const int rows = g.n_pix_tiles * (l.to >= 128 ? kNW / 2 : kNW);
    if (rows <= g.stat_cap) { p.stat_rows = g.stat_rows; *g.stat_written = rows; }
  }
  p.in_bytes = (int32_t)g.in_bytes; p.w_bytes = (int32_t)g.w_bytes; p.a_bias = g.a_bias;
  p.IH = g.IH; p.IW = g.IW; p.IC = g.IC; p.ICp = g.ICp; p.csteps = g.ICp / kBK; p.OH = g.OH; p.OW = g.OW; p.row_pitch = g.row_pitch;
  p.a = g.a; p.b = g.b;
  p.in_cstride = g.in_cstride; p.in_coff = g.in_coff; p.out_cstride = g.out_cstride; p.out_coff = g.out_coff;
  p.act = g.act; p.slope = g.slope; p.mask = g.mask; p.mask_slope = g.mask_slope;
  p.n_oc_tiles = g.n_oc_tiles; p.n_tiles = g.n_pix_tiles * g.n_oc_tiles; p.n_classes = g.tt.n_classes; p.cls_interleave = g.cls_interleave;
  ws_div((uint32_t)p.n_oc_tiles, &p.oct_mul, &p.oct_shift);
  ws_div((uint32_t)p.n_classes, &p.ncls_mul, &p.ncls_shift);
  for (int c = 0; c < g.tt.n_classes; c++) {
    WsClass& k = p.cls[c];
    k.tile_start = g.tt.cls_tile_start[c];
    k.Mc = g.N * g.tt.cls_oh[c] * g.tt.cls_ow[c];
    k.oh_ow = (uint32_t)(uint16_t)g.tt.cls_oh[c] | ((uint32_t)(uint16_t)g.tt.cls_ow[c] << 16);
    k.ph_pw_taps = (uint32_t)(uint8_t)g.tt.cls_ph[c] | ((uint32_t)(uint8_t)g.tt.cls_pw[c] << 8) |
                   ((uint32_t)g.tt.cls_tap_start[c] << 16) | ((uint32_t)g.tt.cls_tap_start[c + 1] << 24);
    ws_div((uint32_t)g.tt.cls_ow[c], &k.ow_mul, &k.ow_shift);
    ws_div((uint32_t)g.tt.cls_oh[c], &k.oh_mul, &k.oh_shift);
  }
  for (int e = 0; e < g.tt.cls_tap_start[g.tt.n_classes]; e++)
    p.taps[e] = ((int)g.tt.dh[e] & 0xff) | (((int)g.tt.dw[e] & 0xff) << 8) | ((int)g.tt.wtap[e] << 16);

  // Workgroups: as many as stay resident (the ring's LDS decides), each walking ceil(tiles / grid) tiles; the grid is
  // trimmed so that every workgroup gets the same number of tiles where that is possible.
  static const int force_stages = getenv("GC_WS_STAGES") ? atoi(getenv("GC_WS_STAGES")) : 0;
  static const int force_slots = getenv("GC_WS_SLOTS") ? atoi(getenv("GC_WS_SLOTS")) : 0;
  // 128-channel tiles: two ring buffers leave room for two workgroups per CU; with at most one tile per CU a second
  // workgroup has nothing to do and a deeper ring hides more of the load latency instead
  static const int deep_ring = getenv("GC_WS_DEEP_RING") ? atoi(getenv("GC_WS_DEEP_RING")) : 1;
  const int stages = (force_stages && p.mask == nullptr) ? force_stages
                     : (l.to == 128 ? ((deep_ring && p.mask == nullptr && p.n_tiles <= 256) ? 3 : 2) : 3);
  const int lds = stages * (l.to + kTP) * 128 + kMaxTaps * 4;
  int per_cu = (160 * 1024) / lds;
  if (per_cu > 2) per_cu = 2;
  if (per_cu < 1) per_cu = 1;
  const int slots = force_slots ? force_slots : device_cu_count() * per_cu;
  const int rounds = (p.n_tiles + slots - 1) / slots;
  const int grid = (p.n_tiles + rounds - 1) / rounds;
  // Few tiles with long reductions (512-channel layers on 8 x 8 images and smaller: a handful of workgroups walking 128
  // steps each on a 256-CU chip): deal each tile's steps to several workgroups when the caller offered scratch memory.
  if (p.mask == nullptr && g.split_ws != nullptr && !force_stages && p.n_tiles < 128) {
    int total = -1;
    bool same = true;
    for (int c = 0; c < g.tt.n_classes; c++) {
      const int st = (g.tt.cls_tap_start[c + 1] - g.tt.cls_tap_start[c]) * p.csteps;
      if (total >= 0 && st != total) same = false;
      total = st;
    }
    long want = (512 + p.n_tiles - 1) / p.n_tiles;
    if (want > total / 8) want = total / 8;                // at least 8 steps per part
    if (want > 32) want = 32;
    const size_t unit_bytes = (size_t)l.to * kTP * sizeof(float);
    const long fit = (long)(g.split_ws_bytes / (unit_bytes * (size_t)p.n_tiles));
    if (want > fit) want = fit;
    if (same && want > 1 && (reinterpret_cast<uintptr_t>(g.split_ws) & 15u) == 0) {
      WsParams q = p;
      q.steps_per_split = (total + (int)want - 1) / (int)want;
      q.ksplit = (total + q.steps_per_split - 1) / q.steps_per_split;
      q.nsum = 0;
      q.part = reinterpret_cast<float*>(g.split_ws);
      ws_div((uint32_t)q.ksplit, &q.ks_mul, &q.ks_shift);
      q.stat_rows = nullptr;                               // (statistics belong to the launch that stores the output)
      const int units = q.n_tiles * q.ksplit;
      const int r1 = (units + slots - 1) / slots, grid1 = (units + r1 - 1) / r1;
      WsParams a = p;
      a.ksplit = 1; a.nsum = q.ksplit; a.part = q.part; a.steps_per_split = 0;
      ws_div(1u, &a.ks_mul, &a.ks_shift);
      int rc;
#define WS_SPLIT(TO_, ST_)                                                                                                        \
      if (l.to == TO_) {                                                                                                            \
        rc = out_f32 ? launch_one<float, TO_, ST_, false, true>(q, grid1, l.flops, stream)                                          \
                     : launch_one<uint16_t, TO_, ST_, false, true>(q, grid1, l.flops, stream);                                      \
        if (rc != GC_OK) return rc;                                                                                                 \
        return out_f32 ? launch_one<float, TO_, ST_, false, true>(a, grid, 0.0, stream)                                             \
                       : launch_one<uint16_t, TO_, ST_, false, true>(a, grid, 0.0, stream);                                         \
      }
      WS_SPLIT(64, 3) WS_SPLIT(128, 2)
#undef WS_SPLIT
    }
  }
  if (p.mask != nullptr) {       // masked variants exist for the default ring depths only
    if (l.to == 64) return out_f32 ? launch_one<float, 64, 3, true>(p, grid, l.flops, stream) : launch_one<uint16_t, 64, 3, true>(p, grid, l.flops, stream);
    return out_f32 ? launch_one<float, 128, 2, true>(p, grid, l.flops, stream) : launch_one<uint16_t, 128, 2, true>(p, grid, l.flops, stream);
  }
#define WS_CASE(TO_, ST_)                                                                                       \
  if (l.to == TO_ && stages == ST_)                                                                             \
    return out_f32 ? launch_one<float, TO_, ST_>(p, grid, l.flops, stream) : launch_one<uint16_t, TO_, ST_>(p, grid, l.flops, stream);
  WS_CASE(64, 2) WS_CASE(64, 3) WS_CASE(64, 4) WS_CASE(128, 2) WS_CASE(128, 3) WS_CASE(128, 4)
#undef WS_CASE
  set_error("internal: no wave-specialised variant for this tile");
  return GC_UNSUPPORTED_GEOMETRY;
}

}  // namespace gc
