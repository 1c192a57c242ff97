/*
 * CPU oracle for the convolution hot path.  TEST INFRASTRUCTURE ONLY: nothing the product ships may
 * import, link or call this file; it exists so that tests, the smoke check and the benchmark's CPU-baseline
 * leg can check the HIP kernels against an independent restatement.
 *
 * The reference repository (mishazakharov/GAN) has no arithmetic of its own on this path: every
 * convolution is a call into PyTorch (torch 2.10.0 here; the reference pins no version), whose published
 * semantics this file restates in plain C loops with double-precision accumulation:
 *   - torch.nn.Conv2d / F.conv2d            : cross-correlation, zero padding,
 *                                             Ho = floor((H + 2p - d(k-1) - 1)/s) + 1
 *   - torch.nn.ConvTranspose2d / F.conv_transpose2d : the gradient of Conv2d w.r.t. its input,
 *                                             Ho = (H-1)s - 2p + d(k-1) + output_padding + 1,
 *                                             weight laid out [Cin, Cout/groups, kh, kw]
 * Reference call sites restated here (paths relative to the reference checkout):
 *   BasicGANs/models.py:35   ConvTranspose2d(in, out, 4, stride, padding, bias=False)
 *   BasicGANs/models.py:77   Conv2d(in, out, 4, stride, padding, bias=False)
 *   Pix2Pix/models.py:83,123 Conv2d / ConvTranspose2d k4 s2 p1 with bias
 *   StyleGAN/model.py:72     Conv2d via EqualizedConv2d (3x3 p1, 1x1, 4x4 s2 p1, 4x4 p0)
 *   StyleGAN/model.py:95,102,120  F.conv2d(x, blur_kernel, padding=1, groups=C)  (depthwise 3x3)
 *   StyleGAN/model.py:404    F.conv_transpose2d(x, w, bias, stride=2, padding=1)
 *   BigGAN/model.py:56-58    F.conv2d(x, W/sigma, bias, stride, padding, dilation, groups)
 * Backward passes follow autograd's definition (aten::convolution_backward): dgrad is the adjoint of the
 * forward map in x, wgrad the adjoint in w, bias grad the sum of gy over N,Ho,Wo.
 *
 * Pinning: tests/test_oracle_golden.py checks every function here against tests/golden/ vectors that were
 * produced by running the reference's own model files (imported by path) on the CPU of the build
 * container (tests/golden/make_golden.py).
 *
 * All tensors are dense NCHW float32; weights OIHW (conv) or IOHW (transposed conv).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX4(a, b, c, d, B, C, D) ((((size_t)(a) * (B) + (b)) * (C) + (c)) * (D) + (d))

/* y[n,co,ho,wo] = bias[co] + sum_{ci,kh,kw} x[n, g*Cg+ci, ho*s-p+kh*d, wo*s-p+kw*d] * w[co,ci,kh,kw] */
void oracle_conv2d_fprop(const float* x, const float* w, const float* bias, float* y,
                         int N, int Cin, int H, int W, int Cout, int kh, int kw,
                         int sh, int sw, int ph, int pw, int dh, int dw, int groups, int Ho, int Wo) {
  int Cg = Cin / groups, Og = Cout / groups;
  for (int n = 0; n < N; n++)
    for (int co = 0; co < Cout; co++) {
      int g = co / Og;
      for (int ho = 0; ho < Ho; ho++)
        for (int wo = 0; wo < Wo; wo++) {
          double acc = bias ? (double)bias[co] : 0.0;
          for (int ci = 0; ci < Cg; ci++)
            for (int r = 0; r < kh; r++) {
              int hi = ho * sh - ph + r * dh;
              if (hi < 0 || hi >= H) continue;
              for (int s = 0; s < kw; s++) {
                int wi = wo * sw - pw + s * dw;
                if (wi < 0 || wi >= W) continue;
                acc += (double)x[IDX4(n, g * Cg + ci, hi, wi, Cin, H, W)] *
                       (double)w[IDX4(co, ci, r, s, Cg, kh, kw)];
              }
            }
          y[IDX4(n, co, ho, wo, Cout, Ho, Wo)] = (float)acc;
        }
    }
}

/* gx = adjoint of fprop in x:  gx[n,ci,hi,wi] = sum over (co,r,s,ho,wo) with hi = ho*s-p+r*d of gy*w.
 * Written as a gather over the output pixel so it is independent of the scatter form used below. */
void oracle_conv2d_dgrad(const float* gy, const float* w, float* gx,
                         int N, int Cin, int H, int W, int Cout, int kh, int kw,
                         int sh, int sw, int ph, int pw, int dh, int dw, int groups, int Ho, int Wo) {
  int Cg = Cin / groups, Og = Cout / groups;
  for (int n = 0; n < N; n++)
    for (int c = 0; c < Cin; c++) {
      int g = c / Cg, ci = c % Cg;
      for (int hi = 0; hi < H; hi++)
        for (int wi = 0; wi < W; wi++) {
          double acc = 0.0;
          for (int r = 0; r < kh; r++) {
            int t = hi + ph - r * dh;
            if (t < 0 || t % sh) continue;
            int ho = t / sh;
            if (ho >= Ho) continue;
            for (int s = 0; s < kw; s++) {
              int u = wi + pw - s * dw;
              if (u < 0 || u % sw) continue;
              int wo = u / sw;
              if (wo >= Wo) continue;
              for (int o = 0; o < Og; o++) {
                int co = g * Og + o;
                acc += (double)gy[IDX4(n, co, ho, wo, Cout, Ho, Wo)] * (double)w[IDX4(co, ci, r, s, Cg, kh, kw)];
              }
            }
          }
          gx[IDX4(n, c, hi, wi, Cin, H, W)] = (float)acc;
        }
    }
}

/* gw[co,ci,r,s] = sum_{n,ho,wo} gy[n,co,ho,wo] * x[n, g*Cg+ci, ho*s-p+r*d, wo*s-p+s*d];  gb[co] = sum gy */
void oracle_conv2d_wgrad(const float* x, const float* gy, float* gw, float* gb,
                         int N, int Cin, int H, int W, int Cout, int kh, int kw,
                         int sh, int sw, int ph, int pw, int dh, int dw, int groups, int Ho, int Wo) {
  int Cg = Cin / groups, Og = Cout / groups;
  for (int co = 0; co < Cout; co++) {
    int g = co / Og;
    for (int ci = 0; ci < Cg; ci++)
      for (int r = 0; r < kh; r++)
        for (int s = 0; s < kw; s++) {
          double acc = 0.0;
          for (int n = 0; n < N; n++)
            for (int ho = 0; ho < Ho; ho++) {
              int hi = ho * sh - ph + r * dh;
              if (hi < 0 || hi >= H) continue;
              for (int wo = 0; wo < Wo; wo++) {
                int wi = wo * sw - pw + s * dw;
                if (wi < 0 || wi >= W) continue;
                acc += (double)gy[IDX4(n, co, ho, wo, Cout, Ho, Wo)] *
                       (double)x[IDX4(n, g * Cg + ci, hi, wi, Cin, H, W)];
              }
            }
          gw[IDX4(co, ci, r, s, Cg, kh, kw)] = (float)acc;
        }
    if (gb) {
      double acc = 0.0;
      for (int n = 0; n < N; n++)
        for (int i = 0; i < Ho * Wo; i++) acc += (double)gy[((size_t)n * Cout + co) * Ho * Wo + i];
      gb[co] = (float)acc;
    }
  }
}

/* Transposed convolution, scatter form as in the PyTorch documentation (groups = 1):
 * y[n,co,hi*s-p+r*d, wi*s-p+s*d] += x[n,ci,hi,wi] * w[ci,co,r,s];  y starts at bias. */
void oracle_conv_transpose2d_fprop(const float* x, const float* w, const float* bias, float* y,
                                   int N, int Cin, int H, int W, int Cout, int kh, int kw,
                                   int sh, int sw, int ph, int pw, int dh, int dw, int Ho, int Wo) {
  size_t plane = (size_t)Ho * Wo;
  double* acc = (double*)__builtin_malloc(sizeof(double) * plane);
  for (int n = 0; n < N; n++)
    for (int co = 0; co < Cout; co++) {
      for (size_t i = 0; i < plane; i++) acc[i] = bias ? (double)bias[co] : 0.0;
      for (int ci = 0; ci < Cin; ci++)
        for (int hi = 0; hi < H; hi++)
          for (int wi = 0; wi < W; wi++) {
            double xv = (double)x[IDX4(n, ci, hi, wi, Cin, H, W)];
            for (int r = 0; r < kh; r++) {
              int ho = hi * sh - ph + r * dh;
              if (ho < 0 || ho >= Ho) continue;
              for (int s = 0; s < kw; s++) {
                int wo = wi * sw - pw + s * dw;
                if (wo < 0 || wo >= Wo) continue;
                acc[(size_t)ho * Wo + wo] += xv * (double)w[IDX4(ci, co, r, s, Cout, kh, kw)];
              }
            }
          }
      for (size_t i = 0; i < plane; i++) y[((size_t)n * Cout + co) * plane + i] = (float)acc[i];
    }
  __builtin_free(acc);
}

/* Depthwise 3x3 binomial blur ([1 2 1] x [1 2 1]) / 16 with zero padding 1 (StyleGAN/model.py:82-150).
 * The kernel is symmetric, so the backward pass (flipped kernel) is the same map applied to gy. */
void oracle_dwblur3x3(const float* x, float* y, int N, int C, int H, int W) {
  static const double k[3] = {1.0, 2.0, 1.0};
  for (size_t nc = 0; nc < (size_t)N * C; nc++)
    for (int h = 0; h < H; h++)
      for (int w = 0; w < W; w++) {
        double acc = 0.0;
        for (int r = -1; r <= 1; r++) {
          int hh = h + r;
          if (hh < 0 || hh >= H) continue;
          for (int s = -1; s <= 1; s++) {
            int ww = w + s;
            if (ww < 0 || ww >= W) continue;
            acc += k[r + 1] * k[s + 1] * (double)x[(nc * H + hh) * W + ww];
          }
        }
        y[(nc * H + h) * W + w] = (float)(acc / 16.0);
      }
}

/* Round-to-nearest-even float32 -> bfloat16 -> float32, the rounding the bf16 kernels apply to their
 * operands; lets a test feed both sides identically rounded inputs and compare at fp32-accumulation
 * tolerance.  NaN stays NaN. */
void oracle_round_bf16(const float* x, float* y, size_t n) {
  for (size_t i = 0; i < n; i++) {
    uint32_t u;
    memcpy(&u, &x[i], 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) { u |= 0x00400000u; u &= 0xffff0000u; }
    else { u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; }
    memcpy(&y[i], &u, 4);
  }
}
