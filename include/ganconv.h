/*
 * ganconv.h -- C ABI of libganconv.so: convolution forward / data-gradient / weight-gradient passes as
 * hand-written gfx950 (MI355X) kernels.
 *
 * The reference repository has no operator or plugin API of its own; the path this library replaces is
 * reached through PyTorch's module / functional calls.  Each entry point names the reference call sites it
 * serves (paths relative to the reference checkout) and the ATen operator that executes there today.
 *
 * Conventions
 *   - Plain C: POD structs, raw device pointers, byte sizes.  No C++ types, no exceptions cross the boundary.
 *   - Every function returns GC_OK (0) or a negative gc_status; gc_last_error() gives a thread-local message.
 *   - Functions only ENQUEUE work on the given stream (a hipStream_t passed as void*; NULL = the null stream)
 *     and return; they never synchronise, never allocate device memory and keep no reference to any buffer.
 *     The caller owns all buffers including the workspace.
 *   - Activations are NHWC ("channels last"): element (n,h,w,c) of a tensor with C channels lives at
 *     ((n*H + h)*W + w) * cstride + coff + c, where cstride/coff default to C/0 and let a caller read from or
 *     write into a channel slice of a wider buffer (U-Net skip concatenation without a copy).
 *   - A gc_conv_desc always describes the CONVOLUTION  x[N,H,W,Cin] -> y[N,Ho,Wo,Cout]; a transposed
 *     convolution is the data-gradient of that convolution, so its forward pass is gc_conv2d_dgrad with
 *     the transposed conv's input in the role of gy and its output in the role of gx.
 *   - Weights are always indexed logically as w[co][ci][r][s] (co over Cout, ci over Cin of the desc), with
 *     arbitrary element strides: torch.nn.Conv2d.weight [Cout,Cin,kh,kw] is that order directly, and
 *     torch.nn.ConvTranspose2d.weight [Cin_t,Cout_t,kh,kw] is that order too because the transposed conv's
 *     input channels are the mirrored conv's output channels.
 *   - Unsupported geometry is an error (GC_UNSUPPORTED_GEOMETRY); there is no fallback to another backend.
 */
#ifndef GANCONV_H
#define GANCONV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GANCONV_VERSION 1

typedef enum gc_status {
  GC_OK = 0,
  GC_BAD_DESC = -1,
  GC_UNSUPPORTED_GEOMETRY = -2,
  GC_WORKSPACE_TOO_SMALL = -3,
  GC_HIP_ERROR = -4,
  GC_BAD_ARGUMENT = -5
} gc_status;

typedef enum gc_dtype { GC_F32 = 0, GC_BF16 = 1 } gc_dtype;

/* Arithmetic of the contraction.  Both accumulate in fp32 on the bf16 matrix cores.
 *   GC_PREC_BF16  : operands rounded to bf16 (one MFMA per product)
 *   GC_PREC_F32X3 : each fp32 operand split into bf16 hi + lo parts, products hi*hi + hi*lo + lo*hi
 *                   (three MFMAs per product; relative error of a product about 2^-16) */
typedef enum gc_precision { GC_PREC_BF16 = 0, GC_PREC_F32X3 = 1 } gc_precision;

typedef enum gc_act { GC_ACT_NONE = 0, GC_ACT_RELU = 1, GC_ACT_LRELU = 2, GC_ACT_TANH = 3, GC_ACT_SIGMOID = 4 } gc_act;

typedef enum gc_pass { GC_PASS_FPROP = 0, GC_PASS_DGRAD = 1, GC_PASS_WGRAD = 2 } gc_pass;

typedef struct gc_conv_desc {
  int32_t N, H, W, Cin;   /* x  : [N, H,  W,  Cin ] */
  int32_t Ho, Wo, Cout;   /* y  : [N, Ho, Wo, Cout]; must satisfy Ho = (H + 2*pad_h - kh)/stride_h + 1 (floor), or,
                             for a transposed conv with output_padding, H may exceed (Ho-1)*stride_h - 2*pad_h + kh */
  int32_t kh, kw;
  int32_t stride_h, stride_w;
  int32_t pad_h, pad_w;
  int32_t x_dtype;        /* gc_dtype of the x-side tensors (x in fprop/wgrad, gx in dgrad) */
  int32_t y_dtype;        /* gc_dtype of the y-side tensors (y in fprop, gy in dgrad/wgrad) */
  int32_t precision;      /* gc_precision */
  int32_t x_cstride, x_coff;   /* 0,0 = dense (cstride = Cin) */
  int32_t y_cstride, y_coff;   /* 0,0 = dense (cstride = Cout) */
  int32_t reserved[4];    /* must be zero */
} gc_conv_desc;

/* Fused output stage of gc_conv2d_fprop / gc_conv2d_dgrad, applied in this order:
 *   v = acc (+ bias[c]) ; v = act(v) (or v *= the derivative selected by grad_mask) ; store ; optional per-channel
 *   statistics of the STORED value.
 * ch_sum / ch_sumsq, when non-NULL, are float[C] accumulators the kernel ADDS into with atomics (the caller
 * zeroes them): sum and sum of squares over N*H*W per output channel -- the batch statistics BatchNorm2d
 * needs right after the conv (BasicGANs/models.py:37,79; Pix2Pix/models.py:86,127). */
typedef struct gc_epilogue {
  int32_t act;            /* gc_act */
  float slope;            /* LeakyReLU negative slope */
  const float* bias;      /* float[C] or NULL */
  float* ch_sum;          /* float[C] or NULL */
  float* ch_sumsq;        /* float[C] or NULL */
  void* out_bf16;         /* NULL, or where to write a bf16 copy of an fp32 output (same layout and channel slice):
                             the operand image of the convolution that reads this output next in GC_PREC_BF16 mode */
  const void* grad_mask;  /* NULL, or a tensor of the output's dtype, layout and channel slice, 16-byte aligned: the stored
                             value is multiplied by 1 where grad_mask > 0 and by `slope` elsewhere.  For a data gradient
                             that flows on through the ReLU (slope 0) / LeakyReLU whose OUTPUT is grad_mask, i.e. the
                             activation in front of the convolution being differentiated: its derivative is applied in the
                             kernel that computes the gradient instead of by aten::leaky_relu_backward /
                             threshold_backward behind it (BasicGANs/models.py:78, the critic's first LeakyReLU).  Needs
                             act == GC_ACT_NONE, no statistics and no out_bf16. */
  float* stat_rows;       /* NULL, or float[stat_rows_capacity][C][2]: the kernel MAY write, per row of output pixels it
                             chooses (one wave's share of a tile), the pair (sum, sum of squares) of the stored values of
                             every output channel -- plain stores in a fixed layout, so that a fixed-order reduction of the
                             rows gives the batch statistics BatchNorm2d needs right behind the conv, bit-wise reproducibly
                             (gc_batchnorm_act_fwd_rows; BasicGANs/models.py:35-37,77-79, Pix2Pix/models.py:83-86,123-127).
                             Unlike ch_sum / ch_sumsq nothing is zeroed by the caller and no float atomics are used. */
  int32_t stat_rows_capacity;
  int32_t* stat_rows_written; /* host int, required with stat_rows: the number of rows written; 0 when the kernel chosen for
                             this problem does not produce them (the caller then computes the statistics from the output) */
  void* workspace;        /* NULL, or scratch memory (16-byte aligned) the call MAY use: a problem with few output tiles and a
                             long reduction (512-channel layers on 8 x 8 images and smaller, Pix2Pix/models.py:100-118) is then
                             split over more workgroups, whose fp32 partial tiles meet here and are added in a fixed order by a
                             second launch (bit-wise reproducible).  4 * 32 bytes per output element is always enough. */
  size_t workspace_bytes;
  const float* sample_scale;     /* NULL, or float[N][C], 16-byte aligned: the accumulator of (sample n, channel c) is multiplied by
                             it in front of the bias:  v = sample_scale[n][c] * acc + bias[c].  The demodulation factor of a
                             StyleGAN2-style modulated convolution -- an EXTENSION, the reference has none (its StyleGAN is
                             version 1, StyleGAN/model.py:179-261). */
  const float* sample_scale_out; /* NULL, or float[N][C], 16-byte aligned: the stored value is sample_scale_out[n][c] * act(v) --
                             the modulation (style) of the NEXT modulated convolution applied by the producer of its input, so
                             that a chain of modulated convolutions is one launch per layer.  Both need C a multiple of 4 and
                             exclude statistics, grad_mask and out_bf16. */
} gc_epilogue;

int gc_version(void);
const char* gc_last_error(void);

/* Size in bytes of the packed weight image consumed by the given pass (GC_PASS_FPROP or GC_PASS_DGRAD). */
size_t gc_packed_weight_bytes(const gc_conv_desc* desc, int pass);

/* Repack fp32 weights w[co][ci][r][s] (element strides w_strides[4]) into the bf16 (hi[,lo]) image the
 * kernels of `pass` read, multiplied by alpha.  alpha carries the per-layer scalar prologues of the reference:
 * EqualizedLR sqrt(2/fan_in) (StyleGAN/model.py:33-37) and SpectralNorm 1/sigma (BigGAN/model.py:20-25). */
int gc_pack_weight(const gc_conv_desc* desc, int pass, const float* w, const int64_t w_strides[4], float alpha,
                   void* packed, void* stream);

/* gc_pack_weight for n weights in one launch (what a training loop wants right after an optimizer step): descs, passes,
 * ws, alphas (may be NULL: all 1) and packed are arrays of n entries, w_strides holds 4 element strides per entry. */
int gc_pack_weights(int32_t n, const gc_conv_desc* descs, const int32_t* passes, const float* const* ws, const int64_t* w_strides,
                    const float* alphas, void* const* packed, void* stream);

/* gc_pack_weights with a divisor per entry that lives in DEVICE memory: divisors[i] is NULL or points to one float, and
 * image i holds (alphas[i] * w) / *divisors[i], rounded as the tensor expression w / sigma rounds.  For the spectral norm
 * (BigGAN/model.py:20-35: the weight of every forward call is W / sigma(W), sigma a device scalar that the power iteration
 * has just advanced): all images of a network in one launch per forward pass, without the tensors W / sigma. */
int gc_pack_weights_div(int32_t n, const gc_conv_desc* descs, const int32_t* passes, const float* const* ws, const int64_t* w_strides,
                        const float* alphas, const float* const* divisors, void* const* packed, void* stream);

/* y = conv2d(x, w) [+ bias, activation, statistics].  Replaces aten::convolution (forward) behind
 *   torch.nn.Conv2d        BasicGANs/models.py:77; Pix2Pix/models.py:17,83,225-232; StyleGAN/model.py:72
 *   F.conv2d               BigGAN/model.py:56-58
 * and, as the gradient of a transposed conv w.r.t. its input, aten::convolution_backward of
 *   torch.nn.ConvTranspose2d  BasicGANs/models.py:35; Pix2Pix/models.py:123-124; StyleGAN/model.py:404.
 * packed_w comes from gc_pack_weight(desc, GC_PASS_FPROP, ...). */
int gc_conv2d_fprop(const gc_conv_desc* desc, const void* x, const void* packed_w, void* y, const gc_epilogue* ep,
                    void* stream);

/* gx = d(conv2d)/dx applied to gy.  Replaces the input-gradient half of aten::convolution_backward for the
 * Conv2d call sites above, and IS the forward pass of
 *   torch.nn.ConvTranspose2d  BasicGANs/models.py:35; Pix2Pix/models.py:123-124
 *   F.conv_transpose2d        StyleGAN/model.py:404
 * (with the optional epilogue adding the transposed conv's bias / activation / statistics on gx).
 * packed_w comes from gc_pack_weight(desc, GC_PASS_DGRAD, ...). */
int gc_conv2d_dgrad(const gc_conv_desc* desc, const void* gy, const void* packed_w, void* gx, const gc_epilogue* ep,
                    void* stream);

/* Workspace bytes gc_conv2d_wgrad needs for this desc (split-K partial sums). */
size_t gc_wgrad_workspace_bytes(const gc_conv_desc* desc);

/* gw[co][ci][r][s] (fp32, element strides gw_strides[4]) = alpha * sum_{n,ho,wo} gy * x  (+ existing value if
 * accumulate != 0); gbias (float[Cout], may be NULL) = sum_{n,ho,wo} gy.  Replaces the weight-gradient half
 * of aten::convolution_backward for every call site above (for a transposed conv pass its input as gy and
 * its output gradient as x). */
int gc_conv2d_wgrad(const gc_conv_desc* desc, const void* x, const void* gy, float* gw, const int64_t gw_strides[4],
                    float alpha, int accumulate, float* gbias, void* workspace, size_t workspace_bytes, void* stream);
/* 1 when gc_conv2d_wgrad on these operands computes gbias inside the weight-gradient launch (the LDS-DMA and row-streaming
 * kernels multiply the gy fragments they hold by a fragment of ones: partial sums per split, added in split order by the
 * reduction launch, bit-wise reproducible), 0 when a pass of its own over gy would follow (which adds with float atomics).
 * A caller that wants a reproducible bias gradient either way asks here and otherwise sums gy itself (gc_channel_sum). */
int gc_wgrad_sums_bias(const gc_conv_desc* desc, const void* x, const void* gy);

/* gc_conv2d_wgrad for callers that cannot allocate: code under HIP-graph capture, tight serving / training loops and
 * harnesses that forbid hipMalloc (and so cannot provide a workspace of a size known only at call time).  No counterpart
 * in the reference.  Same arguments and results as gc_conv2d_wgrad without the workspace pair: the split-K partial sums
 * go to a scratch array that is part of the library's device image (GC_NOALLOC_SCRATCH_BYTES per GPU, held until the
 * process ends), so the call allocates nothing on the host or the device.  A problem whose partial sums need more than
 * that (gc_wgrad_workspace_bytes(desc) > GC_NOALLOC_SCRATCH_BYTES) is rejected with GC_WORKSPACE_TOO_SMALL.  All calls on
 * one GPU share that GPU's array: calls that overlap in time on one GPU (two streams) race; calls on one stream, or on
 * different GPUs, are fine.  The device of the call is the calling thread's current device, which must be the stream's. */
#define GC_NOALLOC_SCRATCH_BYTES (32u << 20)
int gc_conv2d_wgrad_noalloc(const gc_conv_desc* desc, const void* x, const void* gy, float* gw, const int64_t gw_strides[4],
                            float alpha, int accumulate, float* gbias, void* stream);

/* Diagnostic accessor, no counterpart in the reference: the device address and size of the current device's scratch array
 * that gc_conv2d_wgrad_noalloc uses (looked up with hipGetSymbolAddress once per device; nothing is allocated).  For tests
 * that want to see that the array, and not some other buffer, took a call's partial sums. */
int gc_noalloc_scratch(void** ptr, size_t* bytes);

/* Depthwise 3x3 binomial blur ([1 2 1] x [1 2 1]) / 16, zero padding 1, NHWC, dtype gc_dtype.  The kernel is
 * symmetric, so the same call serves forward and backward (StyleGAN/model.py:82-150: BlurFunction,
 * BlurFunctionBackward; F.conv2d(x, kernel, padding=1, groups=C)). */
int gc_dwblur3x3(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, void* stream);
/* The same blur with the derivative of a ReLU (slope 0) / LeakyReLU applied to what it stores (mask_input = 0):
 *   y = blur(x) * (mask > 0 ? 1 : slope),  mask a tensor of y's shape, dtype and layout,
 * or to what it reads (mask_input != 0):  y = blur(x * (mask > 0 ? 1 : slope)), its adjoint.
 * The blur is self-adjoint, so the first is the backward pass of "activation, then blur" in one sweep: x = the gradient
 * arriving at the blur, mask = the activated tensor the blur read in the forward pass (the critic block's LeakyReLU -> Blur,
 * StyleGAN/model.py:428-441), replacing blur + aten::leaky_relu_backward; the second is that sweep's own backward pass
 * (the gradient penalty's double backward, StyleGAN/train.py:226-238). */
int gc_dwblur3x3_masked(const void* x, void* y, const void* mask, float slope, int mask_input, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t dtype, void* stream);

/* InstanceNorm2d (affine = false, no running statistics, biased variance, eps inside the square root) fused with
 * an optional LeakyReLU, NHWC, all tensors of one gc_dtype, `dtype` ( arithmetic and statistics in fp32), C a multiple of
 * 4; HW = H*W.  Replaces aten::instance_norm + aten::leaky_relu
 * and their first- and second-order backward at
 *   torch.nn.InstanceNorm2d(out_dim), torch.nn.LeakyReLU(0.2)   BasicGANs/models.py:79-80 (the WGAN-GP critic)
 *   torch.nn.InstanceNorm2d(channels) inside AdaIN               StyleGAN/model.py:195 (act = GC_ACT_NONE)
 * fwd:      z = act((x - mean) * rsqrt(var + eps))
 * bwd:      gx = d(loss)/dx given gz = d(loss)/dz              (statistics are recomputed from x)
 * bwd_bwd:  given u = d(loss2)/d(gx):  g_x = d(loss2)/dx through gx,  g_gz = d(loss2)/d(gz)
 * The second-order kernel is what the gradient penalty's .backward() needs (BasicGANs/loss_functions.py:21-28).
 * Every *_bf16 argument may be NULL (and must be with dtype = GC_BF16); otherwise the kernel also writes a bf16 (round-to-nearest-even) copy of that
 * output there, same layout, for the convolution that reads it next in GC_PREC_BF16 mode. */
int gc_instnorm_act_fwd(const void* x, void* z, void* z_bf16, int32_t N, int32_t HW, int32_t C, int32_t dtype, float eps, int32_t act,
                        float slope, void* stream);
int gc_instnorm_act_bwd(const void* x, const void* gz, void* gx, void* gx_bf16, int32_t N, int32_t HW, int32_t C, int32_t dtype,
                        float eps, int32_t act, float slope, void* stream);
int gc_instnorm_act_bwd_bwd(const void* x, const void* gz, const void* u, void* g_x, void* g_gz, void* g_x_bf16, void* g_gz_bf16,
                            int32_t N, int32_t HW, int32_t C, int32_t dtype, float eps, int32_t act, float slope, void* stream);

/* BatchNorm2d fused with an optional ReLU / LeakyReLU, NHWC, x / y / gy / gx of one gc_dtype, `dtype` ( parameters,
 * statistics and arithmetic fp32), C a multiple of 4, P = N*H*W pixels.  Replaces
 * aten::batch_norm (+ its backward) followed by aten::relu / aten::leaky_relu at
 *   torch.nn.BatchNorm2d(out_dim), torch.nn.ReLU(True)          BasicGANs/models.py:37, BasicGANs/DCGAN.py:49
 *   norm_op(out_dim), torch.nn.LeakyReLU(0.2) with BatchNorm2d  BasicGANs/models.py:79-80 (the DCGAN critic)
 *   torch.nn.BatchNorm2d(out_ch) of the U-Net blocks            Pix2Pix/models.py:86, 127 (act = GC_ACT_NONE)
 * training != 0: batch statistics (biased variance for the normalisation), running_mean / running_var updated with
 *   `momentum` (unbiased variance) and *num_batches_tracked incremented, as torch.nn.BatchNorm2d does; the running
 *   pointers may be NULL (track_running_stats = False).  training == 0: the running statistics normalise.
 * gamma / beta may be NULL (affine = False).  save_mean / save_rstd ([C]) receive the statistics used; the backward
 * call takes them back.  save_var (optional, [C]) receives the unbiased batch variance, for a caller that keeps the
 * running statistics itself with gc_batchnorm_update_running (running <- (1 - momentum) running + momentum batch,
 * counter + 1) instead of handing the running buffers to the forward call.  The backward call assumes a training-mode forward (statistics depend on x).
 * workspace: gc_batchnorm_workspace_bytes(P, C) bytes, 16-byte aligned.  *_bf16: optional bf16 copy as above. */
size_t gc_batchnorm_workspace_bytes(int64_t P, int32_t C);
int gc_batchnorm_act_fwd(const void* x, void* y, void* y_bf16, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                         void* workspace, size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, float eps, float momentum,
                         int32_t act, float slope, int32_t training, void* stream);
/* gc_batchnorm_act_fwd in training mode with the batch statistics taken from `n_rows` rows of (sum, sum of squares) per
 * channel that the convolution in front wrote (gc_epilogue.stat_rows) instead of from a sweep over x: two launches. */
int gc_batchnorm_act_fwd_rows(const void* x, void* y, void* y_bf16, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                              const float* stat_rows, int32_t n_rows, void* workspace, size_t workspace_bytes, int64_t P, int32_t C,
                              int32_t dtype, float eps, float momentum, int32_t act, float slope, void* stream);
int gc_batchnorm_act_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_rstd, void* gx, void* gx_bf16, float* ggamma, float* gbeta, void* workspace,
                         size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, int32_t act, float slope, void* stream);

/* Conditional BatchNorm fused with an optional ReLU / LeakyReLU: y = act(gamma[n,c] * xh + beta[n,c]) with xh the
 * batch-normalised input (statistics over all N * HW pixels as above, no affine parameters of its own) and a gain and
 * bias per SAMPLE and channel, float [N, C] each.  Replaces BatchNorm.forward of BigGAN/model.py:61-77 (BatchNorm2d
 * (affine=False), then gamma * . + beta from a Linear layer) together with the ReLU that follows it (:129,133).  The
 * backward pass returns ggamma / gbeta as [N, C] and gx; everything else as in gc_batchnorm_act_fwd / _bwd. */
size_t gc_cond_batchnorm_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int gc_cond_batchnorm_act_fwd(const void* x, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var, void* workspace,
                              size_t workspace_bytes, int32_t N, int32_t HW, int32_t C, int32_t dtype, float eps, float momentum,
                              int32_t act, float slope, int32_t training, void* stream);
/* gc_cond_batchnorm_act_fwd in training mode with the batch statistics taken from the rows the convolution in front wrote
 * (gc_epilogue.stat_rows), as gc_batchnorm_act_fwd_rows does for the plain form: BigGAN/model.py:100-133, the second
 * conditional BatchNorm of a generator block behind its up-sampling convolution. */
int gc_cond_batchnorm_act_fwd_rows(const void* x, void* y, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float* save_mean, float* save_rstd, float* save_var,
                                   const float* stat_rows, int32_t n_rows, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW,
                                   int32_t C, int32_t dtype, float eps, float momentum, int32_t act, float slope, void* stream);
int gc_cond_batchnorm_act_bwd(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_rstd, void* gx, float* ggamma, float* gbeta, void* workspace, size_t workspace_bytes,
                              int32_t N, int32_t HW, int32_t C, int32_t dtype, int32_t act, float slope, void* stream);

/* out[c] = sum over the P pixels of x[p][c] (x NHWC fp32 or bf16, dense): the bias gradient of a convolution with
 * bias=True (Pix2Pix/models.py:17,225-232; StyleGAN/model.py:72; BigGAN/model.py:57), which ATen computes with a
 * reduction kernel of its own.  C a multiple of 4: two launches, fixed summation order, workspace from
 * gc_channel_sum_workspace_bytes; other channel counts: element loads and fp32 atomic adds, no workspace. */
size_t gc_channel_sum_workspace_bytes(int64_t P, int32_t C);
int gc_channel_sum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t P, int32_t C, int32_t dtype, void* stream);

/* Backward of an activation that ran in a convolution's epilogue, together with that convolution's bias gradient:
 *   out = gy * act'(y)  (from the activation's output y),  colsum[c] = sum over pixels of out.
 * Replaces leaky_relu_backward / threshold_backward followed by a sum over (N, H, W) for the reference's
 * Conv2d(bias=True) + ReLU / LeakyReLU pairs (Pix2Pix/models.py:235; StyleGAN/model.py:424-447).  NHWC fp32 or bf16, C a
 * multiple of 4, workspace of gc_channel_sum_workspace_bytes(P, C) bytes.  First order only. */
int gc_act_backward_channel_sum(const void* gy, const void* y, void* out, float* colsum, void* workspace, size_t workspace_bytes,
                                int64_t P, int32_t C, int32_t dtype, int32_t act, float slope, void* stream);

/* BatchNorm over a batch that is sharded across GPUs (one process per GPU).  The reference's torch.nn.DataParallel
 * (StyleGAN/train.py:121-125, BigGAN/train.py:150-154) normalises each replica with its own statistics; these three
 * calls let a data-parallel host reproduce the single-device result of BatchNorm2d (BasicGANs/models.py:37,79) instead:
 *   forward:  gc_batchnorm_stats on the shard -> the host merges (mean, unbiased variance, count) over the ranks ->
 *             gc_batchnorm_act_fwd with training = 0 and the merged mean / biased variance as running statistics;
 *   backward: gc_batchnorm_act_bwd_sums on the shard (ggamma = sum g*xh, gbeta = sum g, g = gy * act') -> the host adds
 *             them over the ranks -> gc_batchnorm_act_bwd_apply with means = [2][C] = (sum g, sum g*xh) / global count.
 * Workspace: gc_batchnorm_workspace_bytes(P, C) with the shard's P. */
int gc_batchnorm_stats(const void* x, float* mean, float* var_unbiased, void* workspace, size_t workspace_bytes, int64_t P, int32_t C,
                       int32_t dtype, void* stream);
int gc_batchnorm_act_bwd_sums(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_rstd, float* ggamma, float* gbeta, void* workspace, size_t workspace_bytes, int64_t P,
                              int32_t C, int32_t dtype, int32_t act, float slope, void* stream);
int gc_batchnorm_act_bwd_apply(const void* x, const void* gy, const float* gamma, const float* beta, const float* save_mean,
                               const float* save_rstd, const float* means, void* gx, void* gx_bf16, int64_t P, int32_t C, int32_t dtype,
                               int32_t act, float slope, void* stream);

int gc_batchnorm_update_running(float* running_mean, float* running_var, int64_t* num_batches_tracked, const float* mean,
                                const float* var_unbiased, int32_t C, float momentum, void* stream);

/* One torch.optim.Adam step (amsgrad = false, weight_decay = 0) for n fp32 tensors in two launches.  Replaces the
 * per-tensor / foreach kernels behind optimizer_D.step() / optimizer_G.step() of the reference's training loops (the
 * optimizers are built at BasicGANs/train.py:79-80).  The arrays are host arrays of device
 * pointers; steps[i] points to tensor i's own step count (one device float, the number of steps that tensor has taken
 * so far, as torch.optim.Adam keeps it per parameter); it is incremented first, so the call can be part of a HIP
 * graph.  Same operation order as torch's single-tensor Adam. */
int gc_adam_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                 const int64_t* numel, float* const* steps, float lr, float beta1, float beta2, float eps, void* stream);

/* The same step that also keeps the exponential moving average of the parameters in the same launch:
 *   ema[i] = ema_weight * ema[i] + (1 - ema_weight) * params[i]   (with the updated params)
 * Replaces optimizer_G.step() followed by moving_average(averaged_G, net_G, weight) (StyleGAN/train.py:71-76, called
 * after each generator step), which reads and writes both parameter sets once more.  ema: host array of device
 * pointers, entries may be null (no average kept for that tensor); ema_weight in [0, 1]. */
int gc_adam_ema_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                     float* const* ema, float ema_weight, const int64_t* numel, float* const* steps, float lr, float beta1,
                     float beta2, float eps, void* stream);

/* StyleGAN's chain behind every synthesis convolution as one kernel per direction (StyleGAN/model.py:179-208,249-261:
 * NoiseInput, LeakyReLU(0.2), InstanceNorm2d, style affine):
 *   out = ys[n,c] * IN(lrelu(x + sf[c] * noise[n,hw])) + yb[n,c]
 * x / out / gout / gx NHWC fp32 or bf16 (`dtype`), C a multiple of 4; noise float [N, HW]; sf float [C]; ys, yb float
 * [N, C].  The forward pass writes the per-(n, c) mean and rsqrt(var + eps) of the activated tensor to mu / r
 * (float [N, C]) for the backward pass, which returns gx, g_ys / g_yb [N, C] and g_sf per sample [N, C] (sum over n
 * for the parameter's gradient).  First order only; the noise gets no gradient.  Large images are processed in slabs of
 * pixels (partial statistics, then merge + apply) and need the caller's workspace (16-byte aligned, may be NULL when
 * the size query returns 0). */
size_t gc_noise_lrelu_adain_workspace_bytes(int32_t N, int32_t HW, int32_t C);   /* 0 for small images (one launch, no workspace) */
int gc_noise_lrelu_adain_fwd(const void* x, const float* noise, const float* sf, const float* ys, const float* yb, void* out, float* mu,
                             float* r, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW, int32_t C, int32_t dtype,
                             float eps, float slope, void* stream);
int gc_noise_lrelu_adain_bwd(const void* x, const float* noise, const float* sf, const float* ys, const float* yb, const float* mu,
                             const float* r, const void* gout, void* gx, float* g_sf, float* g_ys, float* g_yb, void* workspace,
                             size_t workspace_bytes, int32_t N, int32_t HW, int32_t C, int32_t dtype, float slope, void* stream);

/* torch.nn.ReflectionPad2d(pad) on NHWC tensors and its adjoint (the gradient of the padding), fp32 or bf16, C a multiple
 * of 8, 16-byte aligned, pad < H and pad < W.  Reference: the reflection-padded 7x7 / 3x3 convolutions of the Pix2PixHD
 * generator (Pix2Pix/models.py:162-163, 185-186, 204-205, 210-211).  H, W are the UNPADDED extents in both calls:
 *   gc_reflect_pad         : x [N, H, W, C] -> y [N, H + 2 pad, W + 2 pad, C]
 *   gc_reflect_pad_adjoint : gy [N, H + 2 pad, W + 2 pad, C] -> gx [N, H, W, C] */
int gc_reflect_pad(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t dtype, void* stream);
int gc_reflect_pad_adjoint(const void* gy, void* gx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t pad, int32_t dtype, void* stream);

/* 2x2 sum-pooling and 2x nearest replication, NHWC, fp32 or bf16 (`dtype`), C a multiple of 8, 16-byte aligned:
 *   gc_pool2x2     : y[n,h,w,c] = scale * (x[n,2h,2w,c] + x[n,2h,2w+1,c] + x[n,2h+1,2w,c] + x[n,2h+1,2w+1,c]),  x is [N,2Ho,2Wo,C]
 *   gc_replicate2x : y[n,2h+a,2w+b,c] = scale * x[n,h,w,c] (a, b in {0,1}),  x is [N,H,W,C], y [N,2H,2W,C]
 * The two are adjoint: AvgPool2d(2) is pool with scale 1/4 and its gradient replicate with 1/4; nearest
 * Upsample(scale_factor=2) is replicate with 1 and its gradient pool with 1.  Replace aten::avg_pool2d(_backward) and
 * aten::upsample_nearest2d(_backward) at BigGAN/model.py:117,188 and StyleGAN/model.py:247,440. */
int gc_pool2x2(const void* x, void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, float scale, void* stream);
int gc_replicate2x(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, float scale, void* stream);
/* The same sweeps with the residual sum of a block in them (BigGAN/model.py:131-133 `h + residual`, :196-200 after the
 * pooling): y = scale * pool(x) + addend  and  y = scale * replicate(x) + addend, `addend` of y's shape and dtype; one
 * rounding of the sum instead of two. */
int gc_pool2x2_add(const void* x, const void* addend, void* y, int32_t N, int32_t Ho, int32_t Wo, int32_t C, int32_t dtype, float scale,
                   void* stream);
int gc_replicate2x_add(const void* x, const void* addend, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype, float scale,
                       void* stream);

/* Channel concatenation of two NHWC tensors of one gc_dtype fused with the (Leaky)ReLU behind it, and its adjoint:
 *   gc_act_concat_fwd   y[p, 0..Ca) = act(a[p, :]), y[p, Ca..Ca+Cb) = act(b[p, :]),  act(v) = v > 0 ? v : slope * v
 *   gc_act_concat_bwd   ga = g[p, 0..Ca) * m, gb = g[p, Ca..Ca+Cb) * m,              m = y > 0 ? 1 : slope   (y: the forward output)
 * P = N * H * W pixels; Ca, Cb multiples of 4 (fp32) / 8 (bf16); 16-byte aligned pointers; slope = 0 is ReLU, 1 a plain
 * concatenation / split.  Replace torch.cat + ReLU (and threshold_backward + the two slices of cat's backward) on the
 * skip connections of the U-Net, Pix2Pix/models.py:128-145 with the up block's ReLU of models.py:30-32. */
int gc_act_concat_fwd(const void* a, const void* b, void* y, int64_t P, int32_t Ca, int32_t Cb, int32_t dtype, float slope, void* stream);
int gc_act_concat_bwd(const void* g, const void* y, void* ga, void* gb, int64_t P, int32_t Ca, int32_t Cb, int32_t dtype, float slope,
                      void* stream);

/* Weight-space prologue of the up-sampling convolutions: "nearest x2, then 3 x 3 convolution" as ONE 4 x 4 stride-2
 * transposed convolution of the small input.  w (fp32) holds n0 * n1 filters of 3 x 3, filter (i0, i1) at
 * w + i0 * stride0 + i1 * stride1 (elements; r, s contiguous), so a [Cout][Cin][3][3] tensor is read as its [Cin][Cout]
 * transpose by swapping the strides; w_out is dense [n0][n1][4][4]:
 *   w_out[a][b] = mult * sum_{r,s} M[a][r] M[b][s] w[r][s]
 *   fold 0: M = [[1 0 0] [1 1 0] [0 1 1] [0 0 1]] / 2  -- FusedUpsample's pad + four-shift average (StyleGAN/model.py:396-402)
 *   fold 1: M = [[0 0 1] [0 1 1] [1 1 0] [1 0 0]]      -- exact fold of interpolate(scale_factor=2) + conv 3x3 padding 1
 *                                                         (BigGAN/model.py:116-120, StyleGAN/model.py:236-238)
 * transpose_grad != 0 runs the adjoint for the gradient: w is then the dense [n0][n1][4][4] gradient and w_out the 3 x 3
 * side (strides as above), w_out[r][s] = mult * sum_{a,b} M[a][r] M[b][s] w[a][b].  One launch each way, in place of the
 * pad / shift / add (or einsum) chain and its autograd graph. */
int gc_fused_upsample_weight(const float* w, float* w_out, int32_t n0, int32_t n1, int64_t stride0, int64_t stride1, float mult,
                             int transpose_grad, int fold, void* stream);

/* Scratch memory a pass of this problem can use: GC_PASS_WGRAD = gc_wgrad_workspace_bytes (required by gc_conv2d_wgrad);
 * GC_PASS_FPROP / GC_PASS_DGRAD = what gc_epilogue.workspace may offer for the optional split of a long reduction (0 for
 * an empty problem).  One query for all three passes. */
size_t gc_workspace_bytes(const gc_conv_desc* desc, int pass);

/* EXTENSION (no counterpart in the reference, whose StyleGAN is version 1 with AdaIN: StyleGAN/model.py:179-261): the two
 * memory-bound halves of a StyleGAN2-style modulated convolution written without per-sample weights,
 *   y[n, o] = d[n, o] * conv(x[n, i] * s[n, i], w)[o],  d[n, o] = rsqrt(sum_i s[n, i]^2 * sum_k w[o, i, k]^2 + eps),
 * i.e. gc_channel_scale on the input, gc_conv2d_fprop, gc_channel_scale on the output.  NHWC tensors of one gc_dtype,
 * C a multiple of 4, s / out float [N, C].
 *   gc_channel_scale   y[n, p, c] = x[n, p, c] * s[n, c]                        (its gradient by x: the same call on gy)
 *   gc_rowdot          out[n, c]  = sum over p of a[n, p, c] * b[n, p, c]       (the gradient by s: a = gy, b = x)
 * gc_rowdot is a fixed-order reduction (no atomics); its workspace holds one partial row per slab of a sample. */
int gc_channel_scale(const void* x, const float* s, void* y, int32_t N, int32_t HW, int32_t C, int32_t dtype, void* stream);
size_t gc_rowdot_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int gc_rowdot(const void* a, const void* b, float* out, void* workspace, size_t workspace_bytes, int32_t N, int32_t HW, int32_t C,
              int32_t dtype, void* stream);

/* EXTENSION, as above: the backward sweep of the fused form, where gc_epilogue's sample_scale (pre) and sample_scale_out
 * (post) put both per-sample scales into the convolution's output stage,  out = post * act(pre * conv + bias)  with act
 * none (slope 1), ReLU (slope 0) or LeakyReLU.  From the incoming gradient g and the STORED out (nothing else is kept from
 * the forward pass: act'(u) * u = act(u) for these activations), one pass writes
 *   gc[n, p, c] = pre[n, c] * m * post[n, c] * g[n, p, c]     m = 1 where post * out > 0, else slope: the gradient of the
 *                                                             convolution's raw output, input of its dgrad / wgrad
 *   dot[n, c]   = sum over p of g * out                       d L / d post = dot / post
 *   sum[n, c]   = sum over p of m * post * g                  d L / d bias = sum over n;  d L / d pre = (dot - bias * sum) / pre
 * pre / post may be NULL (= 1).  Fixed-order reductions, workspace as for gc_rowdot. */
size_t gc_modulated_act_backward_workspace_bytes(int32_t N, int32_t HW, int32_t C);
int gc_modulated_act_backward(const void* g, const void* out, const float* pre, const float* post, void* gc, float* dot, float* sum,
                              void* workspace, size_t workspace_bytes, int32_t N, int32_t HW, int32_t C, int32_t dtype, float slope,
                              void* stream);

/* The weight gradient of ONE layer summed over several (x, gy) pairs in one launch: gw (+)= alpha * sum over segments g of
 * wgrad(xs[g], gys[g]), segment g holding Ns[g] samples (1 to 4 segments; desc->N is ignored, everything else in the
 * descriptor is shared).  For the passes of a training phase that all add to the same weight gradient -- in a WGAN-GP
 * critic iteration (BasicGANs/loss_functions.py:7-28) the real + generated pass and the two contributions of the
 * penalty's double backward: one reduction over all their pixels instead of three launches, three split reductions
 * and two additions.  Results equal the sum of the separate calls up to the order of the fp32 partial sums. */
size_t gc_wgrad_multi_workspace_bytes(const gc_conv_desc* desc, int32_t n_seg, const int32_t* Ns);
int gc_conv2d_wgrad_multi(const gc_conv_desc* desc, int32_t n_seg, const void* const* xs, const void* const* gys, const int32_t* Ns,
                          float* gw, const int64_t gw_strides[4], float alpha, int accumulate, void* workspace, size_t workspace_bytes,
                          void* stream);

/* The scalar parts of a WGAN-GP critic iteration, each one or two launches; they replace the chains of elementwise and
 * reduction tensor operations behind BasicGANs/loss_functions.py:7-27.  fp32 unless `dtype` says otherwise; tensors of
 * N samples with `per_sample` (a multiple of 4) contiguous-or-NHWC elements each, 16-byte aligned, any element order
 * (the operations are order-free within a sample as long as all tensors of a call share it).
 *   gc_wgan_mix              out = real * eps[n] + fake * (1 - eps[n])   (:19; two rounded products and a rounded sum)
 *   gc_gradient_penalty_fwd  norms[n] = ||gradient_n||_2, penalty[0] = lambda * mean_n (norms[n] - 1)^2   (:26-27), and
 *                            coef[n] = lambda * 2 (norms[n] - 1) / (N norms[n]) (0 where the norm is 0), the factor of
 *                            gradient_n in d penalty / d gradient; workspace: gc_gradient_penalty_workspace_bytes
 *   gc_gradient_penalty_bwd  out = g_penalty[0] * coef[n] * gradient_n (same dtype as gradient)
 *   gc_wgan_critic_loss      scores: B real-batch scores then B generated-batch scores.  out[0] = mean(real) - drift *
 *                            mean(real^2) (:7-8, drift 0.001), out[1] = mean(generated) (:13-14), grad [2B] = gradient of
 *                            (out[1] - out[0]) with respect to the scores, i.e. of the two backward passes :9 and :15 */
int gc_wgan_mix(const float* real, const float* fake, const float* eps, float* out, int32_t N, int64_t per_sample, void* stream);
size_t gc_gradient_penalty_workspace_bytes(int32_t N, int64_t per_sample);
int gc_gradient_penalty_fwd(const void* gradient, int32_t dtype, int32_t N, int64_t per_sample, float lambda, float* norms, float* coef,
                            float* penalty, void* workspace, size_t workspace_bytes, void* stream);
int gc_gradient_penalty_bwd(const void* gradient, int32_t dtype, int32_t N, int64_t per_sample, const float* coef, const float* g_penalty,
                            void* out, void* stream);
int gc_wgan_critic_loss(const float* scores, int32_t B, float drift, float* out, float* grad, void* stream);

/* Input side of the step: a batch of decoded 8-bit images (interleaved channels, rows `src_stride` bytes apart, in
 * device memory) to the float batch the networks take.  Per sample, in the order of the reference's datasets
 * (BasicGANs/utils.py:46-55; Pix2Pix/dataset.py:225-318; StyleGAN/train.py:30-38): cv2.resize to resize_w x resize_h
 * (interp 1: INTER_LINEAR with OpenCV's 8-bit fixed-point arithmetic, 0: INTER_NEAREST, the label masks), the window
 * [crop_y, crop_y + out_h) x [crop_x, crop_x + out_w) of the resized image (random jitter), cv2.flip(., 1) of that window
 * when `flip`, value / divisor (255 for images, 1 for label masks), (value - mean[c]) / std[c] (host arrays of `channels`
 * floats, or null: 0 / 1), stored at out + n * s[0] + c * s[1] + y * s[2] + x * s[3] (elements; any layout).  `jobs`: n
 * records in DEVICE memory.  Channels stay in the file's order (cv2: BGR), as in the reference.  out_dtype: GC_F32 / GC_BF16. */
typedef struct {
  const void* src;                     /* first byte of the image */
  int32_t src_h, src_w, src_stride;    /* rows, columns, bytes per row */
  int32_t resize_h, resize_w;          /* size after cv2.resize */
  int32_t crop_y, crop_x;              /* window of the resized image that becomes the output */
  int32_t flip;                        /* 1: mirror the window left-right */
} gc_image_job;
int gc_image_batch_prepare(const gc_image_job* jobs, int32_t n, int32_t channels, int32_t out_h, int32_t out_w, int32_t interp,
                           float divisor, const float* mean, const float* std_, void* out, int32_t out_dtype,
                           const int64_t out_strides[4], void* stream);

/* Spectral norm of a weight by power iteration, as SpectralNorm._power_iterations computes it (BigGAN/model.py:27-35,
 * p = 2): per iteration v = normalize(u W), u' = normalize(v W^T), sigma = u' W v^T, for W [rows][cols] row-major fp32 (a
 * convolution or Linear weight flattened to [out, rest]) and the stored left vector u [rows] (U_vector), which is
 * advanced in place; sigma [1] receives the last iteration's value.  normalize() divides by max(norm, 1e-12) as
 * torch.nn.functional.normalize does.  Fixed summation order.  workspace: gc_spectral_norm_workspace_bytes, 16-byte aligned. */
size_t gc_spectral_norm_workspace_bytes(int32_t rows, int64_t cols);
int gc_spectral_norm_power_iteration(const float* w, int32_t rows, int64_t cols, float* u, float* sigma, int32_t n_iterations,
                                     void* workspace, size_t workspace_bytes, void* stream);
/* The same for every spectrally normalised weight of a network in one set of four launches per iteration (the
 * reference runs the iteration inside each layer's forward call, BigGAN/model.py:22-25; the weights do not change
 * inside a forward pass, so doing all of them in front of it gives the same values).  gc_spectral_norm_multi_plan (host
 * only) turns n jobs into n plan records, the workspace size and the two grid sizes; the caller copies the records to
 * device memory once and calls gc_spectral_norm_multi_run with that copy before every forward pass. */
typedef struct {
  const void* w;                 /* [rows][cols] fp32 */
  void* u;                       /* [rows] fp32, advanced in place */
  void* sigma;                   /* [1] fp32 */
  int32_t rows, pad;
  int64_t cols;
} gc_spectral_job;
typedef struct {
  const void* w;
  void* u;
  void* sigma;
  int64_t cols;
  int64_t ws_part, ws_t, ws_s, ws_scal;      /* offsets into the workspace, in floats */
  int32_t rows, vec, splits, rows_per_split, col_blocks, col_block0, row_block0, pad;
} gc_spectral_plan;
int gc_spectral_norm_multi_plan(const gc_spectral_job* jobs, int32_t n, gc_spectral_plan* plan, size_t* workspace_bytes, int32_t grids[2]);
int gc_spectral_norm_multi_run(const gc_spectral_plan* plan_device, int32_t n, const int32_t grids[2], int32_t n_iterations, void* workspace,
                               void* stream);

/* Self-attention over image positions without the [N, N] weight matrix in memory: out = softmax(k q^T) v per sample,
 * softmax over the last index, unit scale.  Replaces the bmm + Softmax + bmm of SelfAttentionBlock.forward
 * (BigGAN/model.py:97-103; the reference calls the [B, N, C / 8] operand of the first bmm "key" and the other
 * "query": k indexes the softmax rows, q the softmax columns).  All tensors bf16, contiguous, 16-byte aligned:
 * k, q [B, N, d], v / out / dout / dv [B, N, C], dk, dq [B, N, d]; lse and delta fp32 [B, N] (lse is written by the
 * forward pass and read by the backward pass, delta is scratch of the backward pass).  Supported: N a multiple of 128,
 * d 8 or 16, C 64 or 128 (the reference's blocks at ch = 64: 64 channels in the discriminator, 128 in the generator).
 * Accumulation and the softmax statistics are fp32; the backward pass uses no atomics. */
int gc_attention_fwd(const void* k, const void* q, const void* v, void* out, float* lse, int32_t B, int32_t N, int32_t d, int32_t C,
                     void* stream);
int gc_attention_bwd(const void* k, const void* q, const void* v, const void* out, const void* dout, const float* lse, void* dk,
                     void* dq, void* dv, float* delta, int32_t B, int32_t N, int32_t d, int32_t C, void* stream);

/* The tail of the self-attention block (BigGAN/model.py:97-105): out = x + gamma * mixed.view(B, C, H, W), where mixed is
 * the attention product [B, N, C] that the reference reinterprets, WITHOUT a transpose, as a contiguous [B, C, H, W] tensor.
 * With NHWC activations (x, out, g: [B][N][C]) that is a batched transpose fused with an axpy:
 *   fwd   out[b][p][c] = x[b][p][c] + gamma[0] * mixed[b][c * N + p]
 *   bwd   dmixed[b][c * N + p] = gamma[0] * g[b][p][c],  dgamma[0] = sum of g[b][p][c] * mixed[b][c * N + p]   (dx = g)
 * N and C multiples of 64, one gc_dtype for the tensors, gamma / dgamma float[1] on the device, 16-byte aligned pointers;
 * the workspace holds one partial of dgamma per 64 x 64 tile, added in a fixed order. */
int gc_attention_residual_fwd(const void* x, const void* mixed, const float* gamma, void* out, int32_t B, int32_t N, int32_t C, int32_t dtype,
                              void* stream);
size_t gc_attention_residual_workspace_bytes(int32_t B, int32_t N, int32_t C);
int gc_attention_residual_bwd(const void* g, const void* mixed, const float* gamma, void* dmixed, float* dgamma, void* workspace,
                              size_t workspace_bytes, int32_t B, int32_t N, int32_t C, int32_t dtype, void* stream);

/* Split reductions.  A convolution whose output grid cannot fill the chip (few output pixels x channels, long
 * reduction) is split over several workgroups that add fp32 partial tiles into the zeroed output with atomics; the
 * order of those adds, and so the last bit of the result, varies from run to run.  gc_set_split_reduction(0) turns
 * the split off for the whole process (every output element is then produced by one workgroup in a fixed order);
 * the default is on.  Returns the previous setting.  No counterpart in the reference, whose cuDNN algorithms are
 * chosen by torch.backends.cudnn flags; per-channel bias gradients and fused channel statistics are summed with
 * atomics in either setting. */
int gc_set_split_reduction(int on);

/* Kernel selection aid for measurements: gc_set_patch_kernel(0) keeps large stride-1 multi-tap convolutions on the
 * tiled LDS-DMA kernel instead of the patch-resident one (default on; same results up to summation order).  Returns
 * the previous setting.  No counterpart in the reference. */
int gc_set_patch_kernel(int on);

/* Per-kernel timing with HIP events recorded on the launch stream around every launch while enabled.
 * gc_profile_collect synchronises on the recorded events, returns the number of kernel families written to
 * the caller's arrays (at most max_entries) and resets the records.  flops[] holds the algorithmic FLOPs
 * (2 * MACs, necessary taps only) of the launches counted in launches[]. */
int gc_profile_enable(int on);
int gc_profile_collect(int max_entries, char names[][96], int64_t* launches, double* total_ms, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* GANCONV_H */
