"""TEST INFRASTRUCTURE ONLY.  The arithmetic of the library's bf16 precision mode, restated on stock torch CPU ops: every
convolution kernel (forward, data gradient, weight gradient, and the same three again wherever a backward pass is itself
differentiated) rounds its two tensor operands to bf16 (round to nearest even) and accumulates in fp32; its output, the
bias, every other layer and the storage between layers stay fp32.

    with oracle.bf16_operands.conv_operands_rounded():
        loss = restated_model(x)          # oracle/ref_models.py, ref_stylegan.py, ref_biggan.py: untouched
        loss.backward()

Inside the context torch.nn.functional.conv2d / conv_transpose2d (which torch.nn.Conv2d / ConvTranspose2d call) with
groups == 1 become  G(conv(R(x), R(w))) + bias  with two identity-like autograd functions that are each other's backward:

    R: forward rounds to bf16, backward applies G        G: forward is the identity, backward applies R

First order: the incoming gradient passes G's backward = R, so the data- and weight-gradient formulas see the rounded gy
next to the rounded w / x; the gradients leave through R's backward = G unrounded, as the kernels store them in fp32.
Differentiated again (a gradient penalty), the vector arriving at that G is rounded by G's backward = R before it enters
the second-order convolution formulas: the operand rounding of the kernels at every order.  Depthwise convolutions
(groups > 1: StyleGAN's blur, which the library computes in fp32) pass through unchanged.

With this restatement as the comparison a GPU run in bf16 mode differs only by summation order (and by the few weight
folds the harness applies before rounding), so activations fall on the same side of their ReLU / LeakyReLU kinks and
end-to-end gradients can be compared far more sharply than against the fp32 reference traces."""
import contextlib

import torch
import torch.nn.functional as F


class _R(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return _G.apply(g)


class _G(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return _R.apply(g)


def round_operand(t):
    return _R.apply(t)


def round_gradient(t):
    return _G.apply(t)


@contextlib.contextmanager
def conv_operands_rounded():
    conv2d, conv_transpose2d = F.conv2d, F.conv_transpose2d

    def conv2d_r(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1:
            return conv2d(input, weight, bias, stride, padding, dilation, groups)
        y = _G.apply(conv2d(_R.apply(input), _R.apply(weight), None, stride, padding, dilation, 1))
        return y if bias is None else y + bias.view(1, -1, 1, 1)

    def conv_transpose2d_r(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        if groups != 1:
            return conv_transpose2d(input, weight, bias, stride, padding, output_padding, groups, dilation)
        y = _G.apply(conv_transpose2d(_R.apply(input), _R.apply(weight), None, stride, padding, output_padding, 1, dilation))
        return y if bias is None else y + bias.view(1, -1, 1, 1)

    F.conv2d, F.conv_transpose2d = conv2d_r, conv_transpose2d_r
    try:
        yield
    finally:
        F.conv2d, F.conv_transpose2d = conv2d, conv_transpose2d
