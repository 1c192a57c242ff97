"""Drop-in replacements for the two torch.nn.functional calls on the reference's hot path.

    conv2d(x, w, bias, stride, padding, dilation, groups)            BigGAN/model.py:56-58
    conv_transpose2d(x, w, bias, stride, padding, output_padding)    StyleGAN/model.py:404
    upsample2x_conv3x3(x, w, bias)       interpolate(scale_factor=2) + conv2d(3x3, padding 1) as one launch
                                         BigGAN/model.py:116-120, StyleGAN/model.py:236-238

Same argument meaning as torch.nn.functional (ints or pairs).  Geometry the MI355X kernels do not implement
(dilation > 1, groups > 1, anisotropic stride/padding, string padding) raises RuntimeError -- there is no
fallback to another backend.
"""
import os

import torch

from . import ops


def _single(v, name):
    if isinstance(v, (tuple, list)):
        if len(v) == 1:
            return int(v[0])
        if len(v) != 2 or v[0] != v[1]:
            raise RuntimeError(f"ganconv: {name}={tuple(v)} is not supported (must be the same in both dimensions)")
        return int(v[0])
    if isinstance(v, str):
        raise RuntimeError(f"ganconv: {name}='{v}' is not supported")
    return int(v)


def _check_input(x, w):
    if x.dim() != 4 or w.dim() != 4:
        raise RuntimeError(f"ganconv: expected 4-D input and weight, got {tuple(x.shape)} and {tuple(w.shape)}")


_ACT_CODES = {None: 0, "relu": 1, "leaky_relu": 2, "tanh": 3, "sigmoid": 4}


def _act_code(activation):
    if activation not in _ACT_CODES:
        raise RuntimeError(f"ganconv: activation must be one of {list(_ACT_CODES)}")
    return _ACT_CODES[activation]


def _out_code(out_dtype):
    if out_dtype is None:
        return -1
    if out_dtype == torch.float32:
        return 0
    if out_dtype == torch.bfloat16:
        return 1
    raise RuntimeError(f"ganconv: out_dtype must be None, torch.float32 or torch.bfloat16 (got {out_dtype})")


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, *, _wkey=0, precision=None, activation=None,
           negative_slope=0.2, out_dtype=None, weight_scale=1.0, defer_activation_grad=False, input_activation_slope=None):
    """`activation` (None, "relu", "leaky_relu", "tanh", "sigmoid") is applied in the kernel's epilogue: the result
    equals activation(conv2d(...)) of the unfused pair.  `out_dtype` (None = the input's, torch.float32 or
    torch.bfloat16) is the storage type of the result; accumulation is always fp32.  `weight_scale`: a Python scalar c;
    the result is conv2d(input, c * weight, ...) with c folded into the kernel's weight image instead of a separate
    multiplication of the weight on every call.
    `defer_activation_grad` / `input_activation_slope` move the derivative of a ReLU / LeakyReLU epilogue into the
    data-gradient kernel of the next convolution (ops.conv_fprop): the producer is called with defer_activation_grad=True,
    its one consumer with input_activation_slope = the producer's negative slope (0 for ReLU).  Results are those of the
    plain pair; one elementwise sweep over the activated tensor per backward pass goes away."""
    _check_input(input, weight)
    if _single(dilation, "dilation") != 1:
        raise RuntimeError("ganconv: dilation > 1 is not supported")
    if groups != 1:
        raise RuntimeError("ganconv: groups > 1 is not supported by conv2d (the StyleGAN depthwise blur is ops.blur3x3)")
    prec = ops.get_precision() if precision is None else precision
    return ops.conv_fprop(input, weight, bias, _single(stride, "stride"), _single(padding, "padding"), prec, _wkey,
                          _act_code(activation), float(negative_slope), _out_code(out_dtype), float(weight_scale),
                          bool(defer_activation_grad), -1.0 if input_activation_slope is None else float(input_activation_slope))


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1, *, _wkey=0,
                     precision=None, activation=None, negative_slope=0.2, out_dtype=None, weight_scale=1.0):
    """weight is [Cin, Cout, kh, kw] as in torch.  Runs as the data-gradient of the mirrored convolution.
    `activation`, `weight_scale`: as for conv2d."""
    _check_input(input, weight)
    if _single(dilation, "dilation") != 1 or groups != 1:
        raise RuntimeError("ganconv: dilation > 1 / groups > 1 are not supported")
    s, p, op = _single(stride, "stride"), _single(padding, "padding"), _single(output_padding, "output_padding")
    if op >= max(s, 1) and op != 0:
        raise RuntimeError("ganconv: output_padding must be smaller than stride")
    if weight.shape[0] != input.shape[1]:
        raise RuntimeError(f"ganconv: weight expects {weight.shape[0]} input channels, input has {input.shape[1]}")
    kh, kw = weight.shape[2], weight.shape[3]
    H = (input.shape[2] - 1) * s - 2 * p + kh + op
    W = (input.shape[3] - 1) * s - 2 * p + kw + op
    prec = ops.get_precision() if precision is None else precision
    return ops.conv_dgrad(input, weight, bias, H, W, s, p, prec, _wkey, _act_code(activation), float(negative_slope),
                          _out_code(out_dtype), float(weight_scale))


# Nearest x2 replication in front of a 3 x 3 convolution, folded into the weights.  Output row 2i reads the replicated rows
# 2i-1, 2i, 2i+1 = input rows i-1, i, i: taps (w0 | w1 + w2) on rows (i-1 | i); output row 2i+1 reads 2i, 2i+1, 2i+2 =
# input rows i, i, i+1: taps (w0 + w1 | w2) on rows (i | i+1).  That is the tap pattern of a 4-tap stride-2 padding-1
# transposed convolution with k = (w2, w1 + w2, w0 + w1, w0) per dimension, and the zero padding of the replicated image
# coincides with the transposed convolution's missing neighbours at the border.  The fold (and its adjoint for the
# gradient) is one launch of the library's weight-space kernel (ops.upsample_weight_fold, fold 1).


def upsample2x_conv3x3(input, weight, bias=None, *, _wkey=0, precision=None, activation=None, negative_slope=0.2, out_dtype=None,
                       weight_scale=1.0):
    """conv2d(interpolate(input, scale_factor=2, mode="nearest"), weight, bias, stride=1, padding=1) for a [Cout, Cin, 3, 3]
    weight, without the up-scaled tensor: the replication is folded into a [Cin, Cout, 4, 4] weight (sums of the 3 x 3
    taps that fall on the same input pixel) and the result is one stride-2 transposed convolution of the SMALL input:
    4 instead of 9 multiply-adds per output element and channel pair, a quarter of the input traffic, no replication
    launch in either direction.  Same sums as the pair up to the rounding of the added taps (exact for fp32 operands).
    `activation`, `out_dtype`, `weight_scale`: as for conv2d.  Differentiable (to second order) in input, weight, bias."""
    _check_input(input, weight)
    if tuple(weight.shape[2:]) != (3, 3):
        raise RuntimeError(f"ganconv: upsample2x_conv3x3 takes a 3x3 weight, got {tuple(weight.shape[2:])}")
    k4 = ops.upsample_weight_fold(weight, 1, True)          # [Cout, Cin, 3, 3] -> [Cin, Cout, 4, 4]
    return conv_transpose2d(input, k4, bias, 2, 1, _wkey=_wkey, precision=precision, activation=activation,
                            negative_slope=negative_slope, out_dtype=out_dtype, weight_scale=weight_scale)


def modulated_conv2d(input, weight, style, bias=None, stride=1, padding=0, demodulate=True, eps=1e-8, *, _wkey=0, precision=None,
                     activation=None, negative_slope=0.2, out_dtype=None, weight_scale=1.0, input_modulated=False, next_style=None):
    """EXTENSION -- not part of the reference, whose StyleGAN is version 1 (AdaIN, StyleGAN/model.py:179-261).  StyleGAN2's
    modulated convolution: per sample n the weight is w'[o, i, k] = style[n, i] * c * weight[o, i, k] (c = weight_scale)
    and, with `demodulate`, each output channel of w' is normalised to unit L2 norm (plus eps under the root); the result
    is conv2d(input[n], w'[n]) (+ bias, activation).  Computed without per-sample weights, as one ordinary convolution
    whose kernel applies the normalisation factor d[n, o] = rsqrt(sum_i style[n, i]^2 * sum_k (c * weight[o, i, k])^2 + eps)
    to its accumulators in front of bias and activation.  The style scales the INPUT channels: by a sweep over the input
    here, or -- `input_modulated=True` -- by whoever produced the input.  `next_style` ([N, Cout]) asks this call to be that
    producer for the next layer: its output stage multiplies the activated result by next_style[n, o], so a chain of
    modulated convolutions is one launch per layer.  style: [N, Cin].  Differentiable (to second order) in input, weight,
    bias, style and next_style; an entry of next_style that is exactly zero receives a zero gradient."""
    _check_input(input, weight)
    N, Cin, Cout = input.shape[0], input.shape[1], weight.shape[0]
    if style.shape != (N, Cin):
        raise RuntimeError(f"ganconv: style must be [N, Cin] = [{N}, {Cin}], got {tuple(style.shape)}")
    if next_style is not None and next_style.shape != (N, Cout):
        raise RuntimeError(f"ganconv: next_style must be [N, Cout] = [{N}, {Cout}], got {tuple(next_style.shape)}")
    if Cin % 4 or Cout % 4:
        raise RuntimeError("ganconv: modulated_conv2d needs channel counts that are multiples of 4")
    style = style.to(torch.float32)
    x = input if input_modulated else ops.channel_scale(input, style)
    pre = None
    if demodulate:
        w2 = (weight.to(torch.float32) * float(weight_scale)).square().sum((2, 3))          # [Cout, Cin]
        pre = torch.rsqrt(style.square() @ w2.t() + eps)
    code = _act_code(activation)
    prec = ops.get_precision() if precision is None else precision
    fused_act = code if code <= 2 else 0
    y = ops.modconv_fprop(x, weight, bias, pre, next_style if fused_act == code else None, _single(stride, "stride"),
                          _single(padding, "padding"), prec, _wkey, fused_act, float(negative_slope), _out_code(out_dtype),
                          float(weight_scale))
    if fused_act != code:      # tanh / sigmoid: behind the kernel
        y = torch.tanh(y) if code == 3 else torch.sigmoid(y)
        if next_style is not None:
            y = ops.channel_scale(y, next_style.to(torch.float32))
    return y


def blur3x3(input, input_activation_slope=None):
    """Depthwise [1 2 1] x [1 2 1] / 16 blur, zero padding 1.  `input_activation_slope`: as for conv2d -- the input is the
    output of a convolution called with defer_activation_grad=True whose only consumer this blur is; the blur's backward
    sweep applies that activation's derivative."""
    return ops.blur3x3(input, -1.0 if input_activation_slope is None else float(input_activation_slope))


def avg_pool2x2(input):
    """torch.nn.functional.avg_pool2d(input, 2) for even extents; channel counts that are not a multiple of 8 (images)
    take the torch operator."""
    if input.shape[1] % 8 or input.shape[2] % 2 or input.shape[3] % 2:
        return torch.nn.functional.avg_pool2d(input, 2)
    return ops.pool2x2(input, 0.25)


def upsample_nearest2x(input):
    """torch.nn.functional.interpolate(input, scale_factor=2, mode="nearest")."""
    if input.shape[1] % 8:
        return torch.nn.functional.interpolate(input, scale_factor=2, mode="nearest")
    return ops.replicate2x(input, 1.0)


_FUSED_SUMS = os.environ.get("GC_NO_RESAMPLE_ADD") is None


def avg_pool2x2_add(input, addend):
    """avg_pool2x2(input) + addend as one sweep where the fused kernel applies (otherwise the plain pair)."""
    if (not _FUSED_SUMS or input.shape[1] % 8 or input.shape[2] % 2 or input.shape[3] % 2 or addend.dtype != input.dtype
            or not input.is_cuda):
        return avg_pool2x2(input) + addend
    return ops.pool2x2_add(input, addend, 0.25)


def upsample_nearest2x_add(input, addend):
    """upsample_nearest2x(input) + addend as one sweep where the fused kernel applies (otherwise the plain pair)."""
    if not _FUSED_SUMS or input.shape[1] % 8 or addend.dtype != input.dtype or not input.is_cuda:
        return upsample_nearest2x(input) + addend
    return ops.replicate2x_add(input, addend, 1.0)
