// Weight gradient of a 3x3 / stride 1 / padding 1 convolution on bf16 NHWC operands: every filter tap out of ONE copy of x.
//
//   gw[r][s][co][ci] = sum over (n, h, w) of gy[n][h][w][co] * x[n][h + r - 1][w + s - 1][ci]
//
// The tiled kernel (conv_wgrad_dma.hip) treats the nine taps as nine GEMMs: it brings the gy tile into LDS once per tap
// (or pair of taps) and the x tile once per tap, 24 KiB per 128 MFMAs, and sits near the rate at which a CU takes bytes
// in.  Here a workgroup owns a 64 (co) x 64 (ci) block of the gradient for ALL nine taps and walks column strips of the
// images (32 pixels wide, top to bottom).  One reduction step is one image row of a strip: its 32 gy pixels and the x row
// below it are the only new bytes (4 + 4.25 KiB for 144 MFMAs); the x rows h - 1, h, h + 1 stay in a ring in LDS and the
// B operand of tap (r, s) is the transposing read of ring row h + r - 1 shifted by s - 1 pixels.  A wave owns 16 input
// channels x 64 output channels x 9 taps (36 accumulator tiles): per step it reads the four gy fragments once and one x
// fragment per tap, 13 KiB of LDS reads for 36 MFMAs.  Two such workgroups share a CU and drift apart freely.
//
// Loads are LDS-DMA (`buffer_load_dwordx4 ... lds`) through a ring of kR slots, kD slots in flight; slot q of a strip holds
// x row q - 1 (rows -1 and H are zeros: out-of-range lanes of a buffer load return 0) and gy row q - 2, so a strip costs
// two slots without MFMAs.  Rows are 128 bytes per pixel; the 32-byte segment of a pixel is XORed with bits 1 and 3 of
// the pixel's column on the SOURCE side (a DMA instruction writes 1 KiB linearly), which keeps the transposing reads
// (`ds_read_b64_tr_b16`, 16 rows x 32 bytes per half wave) off each other's banks at every shift.
// Partial sums go to the caller's workspace as [split][tap][Cout][Cin] fp32, the layout wgrad_reduce_* sums.
#include <stdio.h>

#include "conv_wgrad.h"

namespace gc {
namespace {

constexpr int kR = 6;                 // ring slots
constexpr int kD = 3;                 // slots in flight behind the one being multiplied
constexpr int kXRow = 5 * 1024;       // 40 pixel slots x 64 channels (34 used: the strip and one pixel on each side)
constexpr int kYRow = 4 * 1024;       // 32 pixels x 64 channels
constexpr int kXBytes = kR * kXRow;
constexpr int kSmem = kR * (kXRow + kYRow);
constexpr uint32_t kOutOfRange = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int N>
__device__ __forceinline__ void wait_vm() {   // s_waitcnt vmcnt(N) only
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (15 << 8));
}

__device__ __forceinline__ int col_swz(int c) { return ((c >> 1) & 1) | (((c >> 3) & 1) << 1); }

__device__ __forceinline__ bf16x8 join(s16x4 lo, s16x4 hi) {
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

#define GC_TR_READ(dst, addr) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr) : "memory")
#define GC_TR_READ_OFF(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")

}  // namespace

// BIAS: the variant that also sums gy (a build of its own, so that the plain one keeps its registers and schedule)
template <bool BIAS>
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_rows_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);

  const uint32_t ntiles = (uint32_t)p.n_co_tiles * (uint32_t)p.n_ci_tiles;
  const uint32_t lid = xcd_remap(blockIdx.x, ntiles * (uint32_t)p.splits);
  const int tile = (int)(lid % ntiles), split = (int)(lid / ntiles);
  const int co0 = (tile / p.n_ci_tiles) * 64, ci0 = (tile % p.n_ci_tiles) * 64;
  const int s_begin = split * p.strips_per_split;
  const int s_end = min(p.total_strips, s_begin + p.strips_per_split);
  const int H = p.H, W = p.W;
  const int spp = H + 2;                                  // slots per strip
  const int nslots = s_end > s_begin ? (s_end - s_begin) * spp : 0;
  const uint32_t x_row_b = (uint32_t)W * (uint32_t)p.x_cstride * 2u, y_row_b = (uint32_t)W * (uint32_t)p.y_cstride * 2u;

  // ---- per-lane source geometry of this wave's DMA instructions (a wave-instruction covers 8 pixel slots of 128 bytes)
  const int l_px = lane >> 3, l_c = lane & 7;
  auto src_chunk = [&](int px) { return ((((l_c >> 1) ^ col_swz(px)) << 1) | (l_c & 1)) * 8; };   // channel of the lane's 16 bytes
  const int xpx0 = wave * 8 + l_px, xpx1 = 32 + l_px;     // x: instruction `wave`, and (wave 0 only) instruction 4
  const int ypx = wave * 8 + l_px;
  const int x_lane0 = ((xpx0 - 1) * p.x_cstride + p.x_coff + ci0 + src_chunk(xpx0)) * 2;
  const int x_lane1 = ((xpx1 - 1) * p.x_cstride + p.x_coff + ci0 + src_chunk(xpx1)) * 2;
  const int y_lane = (ypx * p.y_cstride + p.y_coff + co0 + src_chunk(ypx)) * 2;

  // ---- issue cursor: strip (segment, sample, column strip), slot within the strip, ring slot
  int i_strip = s_begin, i_j = 0, i_ring = 0, i_q = 0;
  int i_seg = 0, i_n = 0, i_ws = 0;
  {
    int rest = s_begin;
#pragma unroll
    for (int g = 1; g < kMaxWgradSegments; g++)
      if (g < p.n_seg && s_begin >= p.seg_split0[g]) i_seg = g;       // (rows mode: seg_split0 = first strip of a segment)
    rest -= p.seg_split0[i_seg];
    i_n = rest / p.strips_w;
    i_ws = rest - i_n * p.strips_w;
  }
  __amdgpu_buffer_rsrc_t x_rs, y_rs;
  uint32_t xv0 = kOutOfRange, xv1 = kOutOfRange, yv = kOutOfRange;    // per-strip lane offsets (row 0 of the strip)
  auto enter_strip = [&]() {
    const void* xp = p.seg_x[0]; const void* yp = p.seg_gy[0];
    int64_t xb = p.seg_x_bytes[0], yb = p.seg_y_bytes[0];
#pragma unroll
    for (int g = 1; g < kMaxWgradSegments; g++)
      if (g == i_seg) { xp = p.seg_x[g]; yp = p.seg_gy[g]; xb = p.seg_x_bytes[g]; yb = p.seg_y_bytes[g]; }
    x_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xp), 0, (int)xb, 0x00020000);
    y_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(yp), 0, (int)yb, 0x00020000);
    const int w0 = i_ws * 32;
    const uint32_t pix0 = (uint32_t)i_n * (uint32_t)H * (uint32_t)W + (uint32_t)w0;
    const uint32_t xs = pix0 * (uint32_t)p.x_cstride * 2u, ys = pix0 * (uint32_t)p.y_cstride * 2u;
    const bool left = w0 == 0, right = w0 + 32 == W;
    xv0 = (xpx0 == 0 && left) ? kOutOfRange : xs + (uint32_t)x_lane0;
    xv1 = (xpx1 > 33 || (xpx1 == 33 && right)) ? kOutOfRange : xs + (uint32_t)x_lane1;
    yv = ys + (uint32_t)y_lane;
  };
  if (nslots > 0) enter_strip();

  auto issue = [&]() {
    char* xdst = smem + i_ring * kXRow;
    char* ydst = smem + kXBytes + i_ring * kYRow;
    const bool live = i_q < nslots;
    const int xr = i_j - 1, yr = i_j - 2;
    const bool x_ok = live && (unsigned)xr < (unsigned)H, y_ok = live && (unsigned)yr < (unsigned)H;
    const uint32_t xo = (uint32_t)xr * x_row_b, yo = (uint32_t)yr * y_row_b;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rs, (lds_ptr_t)(ydst + wave * 1024), 16, (int)(y_ok ? yv + yo : kOutOfRange), 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xdst + wave * 1024), 16, (int)(x_ok ? xv0 + xo : kOutOfRange), 0, 0, 0);
    if (wave == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rs, (lds_ptr_t)(xdst + 4 * 1024), 16, (int)(x_ok ? xv1 + xo : kOutOfRange), 0, 0, 0);
    i_q++;
    i_ring = i_ring + 1 == kR ? 0 : i_ring + 1;
    if (++i_j == spp) {
      i_j = 0;
      if (++i_strip < s_end) {
        if (++i_ws == p.strips_w) {
          i_ws = 0;
          i_n++;
#pragma unroll
          for (int g = 1; g < kMaxWgradSegments; g++)
            if (g < p.n_seg && i_strip == p.seg_split0[g]) { i_seg = g; i_n = 0; }
        }
        enter_strip();
      }
    }
  };

  // ---- fragment addresses.  Transposing read: lane (g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3) addresses row
  // 8g + q (+ 4 for the second read) and 8 bytes at 8 pp inside a 32-byte segment, and receives the values of channel
  // lane & 15 of that segment at rows 8g .. 8g + 3 (+ 4): k = 8g + j of a 16x16x32 MFMA operand.
  const int k0 = 8 * (lane >> 4) + ((lane >> 2) & 3);
  const int sub = (lane & 3) * 8;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  uint32_t ya[4];                      // gy fragment i (output channels 16 i ..), relative to the slot
#pragma unroll
  for (int i = 0; i < 4; i++) ya[i] = (uint32_t)(k0 * 128 + ((i ^ col_swz(k0)) << 5) + sub);
  uint32_t xa[3][2];                   // x fragment of this wave's 16 channels at shift s - 1, rows k0 and k0 + 4
#pragma unroll
  for (int s = 0; s < 3; s++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int c = k0 + 4 * h + s;    // pixel slot: strip pixel k sits in slot k + 1
      xa[s][h] = (uint32_t)(c * 128 + ((wave ^ col_swz(c)) << 5) + sub);
    }

  f32x4 acc[9][4];
#pragma unroll
  for (int a = 0; a < 9; a++)
#pragma unroll
    for (int i = 0; i < 4; i++) acc[a][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient: wave 0 of the workgroups that own input-channel block 0 multiplies the gy fragments it has in registers
  // by a fragment of ones as well, which sums them over the step's 32 pixels (every row of the result holds that sum)
  const bool do_bias = BIAS && p.bias_part != nullptr && ci0 == 0 && wave == 0;
  f32x4 bacc[BIAS ? 4 : 1];
#pragma unroll
  for (int i = 0; i < (BIAS ? 4 : 1); i++) bacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const s16x8 ones_bits = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);

  auto compute = [&](int ring) {
    const uint32_t yb = smem_base + (uint32_t)(kXBytes + ring * kYRow);
    uint32_t xb[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      int rr = ring - 2 + r;
      rr = rr < 0 ? rr + kR : rr;
      xb[r] = smem_base + (uint32_t)(rr * kXRow);
    }
    s16x4 ylo[4], yhi[4], blo[2], bhi[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t ad = yb + ya[i];
      GC_TR_READ(ylo[i], ad);
      GC_TR_READ_OFF(yhi[i], ad, 4 * 128);
    }
    GC_TR_READ(blo[0], xb[0] + xa[0][0]);
    GC_TR_READ(bhi[0], xb[0] + xa[0][1]);
    bf16x8 a[4];
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const int cur = tap & 1, nxt = cur ^ 1;
      if (tap + 1 < 9) {
        GC_TR_READ(blo[nxt], xb[(tap + 1) / 3] + xa[(tap + 1) % 3][0]);
        GC_TR_READ(bhi[nxt], xb[(tap + 1) / 3] + xa[(tap + 1) % 3][1]);
      }
      // the wait names every register it completes, so that no use can be scheduled ahead of it
      if (tap == 0) {
        asm volatile("s_waitcnt lgkmcnt(2)"
                     : "+v"(ylo[0]), "+v"(yhi[0]), "+v"(ylo[1]), "+v"(yhi[1]), "+v"(ylo[2]), "+v"(yhi[2]), "+v"(ylo[3]), "+v"(yhi[3]),
                       "+v"(blo[0]), "+v"(bhi[0]));
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = join(ylo[i], yhi[i]);
      } else if (tap + 1 < 9) {
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(blo[cur]), "+v"(bhi[cur]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo[cur]), "+v"(bhi[cur]));
      }
      const bf16x8 b = join(blo[cur], bhi[cur]);
#pragma unroll
      for (int i = 0; i < 4; i++) acc[tap][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a[i], acc[tap][i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BIAS) {
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; i++) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, a[i], bacc[i], 0, 0, 0);
      }
    }
  };

  // ---- ring: slots q + 1 .. q + kD are in flight while slot q is multiplied; past the last slot the cursor issues
  // loads whose every lane is out of range (zeros into slots nobody reads), so the wait counts stay uniform
  if (nslots > 0) {
#pragma unroll
    for (int s = 0; s < kD; s++) issue();
    int c_j = 0, c_ring = 0;
    for (int q = 0; q < nslots; q++) {
      if (wave == 0) wait_vm<3 * (kD - 1)>(); else wait_vm<2 * (kD - 1)>();
      __builtin_amdgcn_s_barrier();        // slot q has landed for every wave; every wave is done reading slot q - 3
      __builtin_amdgcn_sched_barrier(0);
      issue();                             // slot q + kD into the ring slot of q - 3
      if (c_j >= 2) compute(c_ring);
      c_ring = c_ring + 1 == kR ? 0 : c_ring + 1;
      c_j = c_j + 1 == spp ? 0 : c_j + 1;
      __builtin_amdgcn_sched_barrier(0);
    }
    wait_vm<0>();                          // the zero-filling tail loads target this workgroup's LDS: let them land
  }

  // ---- partial block: accumulator rows are input channels, a lane stores 4 consecutive channels of one output channel
  const int T = 9;
#pragma unroll
  for (int tap = 0; tap < 9; tap++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int co = co0 + i * 16 + (lane & 15);
      const int ci = ci0 + wave * 16 + (lane >> 4) * 4;
      float* dst = p.part + (((size_t)split * T + tap) * p.Cout + co) * p.Cin + ci;
      *reinterpret_cast<f32x4*>(dst) = acc[tap][i];
    }
  if constexpr (BIAS) {
    if (do_bias && lane < 16) {
#pragma unroll
      for (int i = 0; i < 4; i++) p.bias_part[(size_t)split * p.Cout + co0 + i * 16 + lane] = bacc[i][0];
    }
  }
}

// Takes the problem when the geometry is the kernel's: 3x3, stride 1, padding 1, bf16, channel counts multiples of 64,
// image width a multiple of 32, 16-byte aligned channel slices, every tensor below 2 GiB.  Fills the split bookkeeping
// (a split is a run of strips; seg_split0 holds the first strip of each segment).
bool plan_wgrad_rows(WgradParams& p, const WgradLaunch& l, int n_seg, const int32_t* Ns) {
  static const bool off = getenv("GC_NO_WGRAD_ROWS") != nullptr;
  if (off || !l.dma || p.x_f32 || p.kh != 3 || p.kw != 3 || p.sh != 1 || p.sw != 1 || p.ph != 1 || p.pw != 1) return false;
  if (p.Cin % 64 != 0 || p.Cout % 64 != 0 || p.W % 32 != 0 || p.H < 4 || p.Ho != p.H || p.Wo != p.W) return false;
  const int64_t limit = (1ll << 31) - (1 << 20);
  long strips = 0;
  p.strips_w = p.W / 32;
  for (int g = 0; g < n_seg; g++) {
    if ((int64_t)Ns[g] * p.H * p.W * p.x_cstride * 2 >= limit || (int64_t)Ns[g] * p.H * p.W * p.y_cstride * 2 >= limit) return false;
    p.seg_split0[g] = (int32_t)strips;
    strips += (long)Ns[g] * p.strips_w;
  }
  if (strips <= 0 || strips >= (1l << 30)) return false;
  p.n_co_tiles = p.Cout / 64;
  p.n_ci_tiles = p.Cin / 64;
  const long tiles = (long)p.n_co_tiles * p.n_ci_tiles;
  long want = (512 + tiles - 1) / tiles;          // about two workgroups per CU
  if (want > strips) want = strips;
  if (want < 1) want = 1;
  p.strips_per_split = (int32_t)((strips + want - 1) / want);
  p.splits = (int32_t)((strips + p.strips_per_split - 1) / p.strips_per_split);
  // a caller with a bounded workspace (the entry point that allocates nothing): leave the problem to the tiled kernels,
  // whose partial sums are smaller for weights with few tiles, when the strips' partial sums would not fit
  // (the bias partials [splits][Cout] follow the weight partials)
  if (wgrad_rows_workspace_cap() != 0 &&
      (size_t)p.splits * 9 * p.Cout * p.Cin * sizeof(float) + (size_t)p.splits * p.Cout * sizeof(float) + 16 > wgrad_rows_workspace_cap())
    return false;
  p.total_strips = (int32_t)strips;
  p.rows = 1;
  return true;
}

int launch_wgrad_rows(const WgradParams& p, const WgradLaunch& l, hipStream_t stream) {
  if (p.total_strips <= 0 || p.splits <= 0) return GC_OK;
  ProfScope ps("gc::conv_wgrad_rows_kernel", l.flops, stream);
  const unsigned nblocks = (unsigned)(p.n_co_tiles * p.n_ci_tiles * p.splits);
  if (p.bias_part != nullptr) hipLaunchKernelGGL(conv_wgrad_rows_kernel<true>, dim3(nblocks), dim3(kThreads), kSmem, stream, p);
  else hipLaunchKernelGGL(conv_wgrad_rows_kernel<false>, dim3(nblocks), dim3(kThreads), kSmem, stream, p);
  return check_hip(hipGetLastError(), "conv_wgrad_rows launch");
}

}  // namespace gc
