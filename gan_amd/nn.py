"""Drop-in torch.nn.Conv2d / torch.nn.ConvTranspose2d whose forward, data-gradient and weight-gradient passes
run on the hand-written gfx950 kernels.

They SUBCLASS the torch modules: constructor arguments, parameter names/shapes (state_dict compatible with the
reference's checkpoints), `.weight/.bias/.stride/.padding/...` attributes and isinstance checks
(BasicGANs/utils.py:32, BigGAN/train.py:110) all behave as in torch.  `self.weight` is read at call time, so
hooks that replace it with a non-leaf tensor (EqualizedLR, StyleGAN/model.py:40-50) and subclasses that call
the functional form with a transformed weight (BigGAN/model.py:48-58) keep working.

Reference call sites: BasicGANs/models.py:35,77; Pix2Pix/models.py:17,83,123-124,225-232; StyleGAN/model.py:72.
"""
import contextlib
import weakref

import torch

from . import functional as F
from . import ops


_NO_CONTEXT = contextlib.nullcontext()


def fuse_batchnorm_statistics(module, on=True):
    """Marks every Conv2d / ConvTranspose2d of this package that sits right in front of a training-mode BatchNormAct2d
    (non-conditional, statistics per rank) inside a torch.nn.Sequential of `module`: its kernel then also writes the rows of
    per-channel sums the BatchNorm needs, and the BatchNorm skips its own statistics sweep over the tensor
    (BasicGANs/models.py:35-37,77-79; Pix2Pix/models.py:83-86,123-127).  Returns the number of pairs found."""
    n = 0
    for seq in module.modules():
        if isinstance(seq, torch.nn.Sequential):
            kids = list(seq.children())
            for a, b in zip(kids, kids[1:]):
                if isinstance(a, (Conv2d, ConvTranspose2d)) and isinstance(b, BatchNormAct2d) and not getattr(b, "sync_stats", False):
                    a.channel_statistics = bool(on)
                    n += 1
    return n


class _KeyedWeight:
    """Gives a module a packed-weight cache key that is valid while `self.weight` is its own Parameter."""

    def _wkey_for(self, weight):
        if not isinstance(weight, torch.nn.Parameter):
            return 0
        key = self.__dict__.get("_ganconv_wkey")
        if key is None:
            key = ops.new_weight_key()
            self.__dict__["_ganconv_wkey"] = key
            weakref.finalize(self, ops.drop_weight_key, key)
        return key


class _FusedActivation:
    """Optional activation applied inside the convolution kernel: `activation` is None, "relu", "leaky_relu", "tanh"
    or "sigmoid".  The module then stands for the reference's conv + activation pair (the activation module holds no
    state, so state_dicts are unchanged).  `out_dtype` (None = the input's dtype, torch.float32, torch.bfloat16) picks
    the storage type of the output: a network keeps its activations in bf16 by asking for bf16 at the first
    convolution and for float32 at the last; every layer in between follows its input."""

    def _init_activation(self, activation, negative_slope, out_dtype=None):
        F._act_code(activation)
        F._out_code(out_dtype)
        self.activation, self.negative_slope, self.out_dtype = activation, negative_slope, out_dtype

    def extra_repr(self):
        base = super().extra_repr()
        return base if self.activation is None else f"{base}, activation={self.activation}"


class Conv2d(_FusedActivation, _KeyedWeight, torch.nn.Conv2d):
    """defer_activation_grad / input_activation_slope: see functional.conv2d (a producer / consumer pair of convolutions
    with a ReLU or LeakyReLU between them, the consumer being the activated tensor's only reader)."""

    def __init__(self, *args, activation=None, negative_slope=0.2, out_dtype=None, defer_activation_grad=False,
                 input_activation_slope=None, channel_statistics=False, **kwargs):
        super().__init__(*args, **kwargs)
        self._init_activation(activation, negative_slope, out_dtype)
        self.defer_activation_grad, self.input_activation_slope = defer_activation_grad, input_activation_slope
        self.channel_statistics = channel_statistics      # True: a BatchNormAct2d reads the output next (fuse_batchnorm_statistics)

    def _conv_forward(self, input, weight, bias):
        if self.padding_mode != "zeros":
            raise RuntimeError("ganconv: only zero padding is supported")
        key = self._wkey_for(weight) if weight is self._parameters.get("weight") else 0
        with ops.channel_statistics_request() if (self.channel_statistics and self.training) else _NO_CONTEXT:
            return F.conv2d(input, weight, bias, self.stride, self.padding, self.dilation, self.groups, _wkey=key,
                            activation=self.activation, negative_slope=self.negative_slope, out_dtype=self.out_dtype,
                            defer_activation_grad=self.defer_activation_grad, input_activation_slope=self.input_activation_slope)


class ConvTranspose2d(_FusedActivation, _KeyedWeight, torch.nn.ConvTranspose2d):
    def __init__(self, *args, activation=None, negative_slope=0.2, out_dtype=None, channel_statistics=False, **kwargs):
        super().__init__(*args, **kwargs)
        self._init_activation(activation, negative_slope, out_dtype)
        self.channel_statistics = channel_statistics

    def forward(self, input, output_size=None):
        if self.padding_mode != "zeros":
            raise RuntimeError("ganconv: only zero padding is supported")
        if output_size is not None:
            raise RuntimeError("ganconv: output_size is not supported; use output_padding")
        weight = self.weight
        key = self._wkey_for(weight) if weight is self._parameters.get("weight") else 0
        with ops.channel_statistics_request() if (self.channel_statistics and self.training) else _NO_CONTEXT:
            return F.conv_transpose2d(input, weight, self.bias, self.stride, self.padding, self.output_padding, self.groups,
                                      self.dilation, _wkey=key, activation=self.activation, negative_slope=self.negative_slope,
                                      out_dtype=self.out_dtype)


class ModulatedConv2d(_KeyedWeight, torch.nn.Module):
    """EXTENSION (StyleGAN2; the reference has no counterpart): convolution whose weight is modulated per sample by a
    style vector and optionally demodulated, see functional.modulated_conv2d.  `affine` maps the latent [N, style_dim] to
    the per-input-channel scale (bias initialised to 1, as in StyleGAN2); weight [Cout, Cin, k, k] ~ N(0, 1) with the
    equalised-learning-rate scalar 1 / sqrt(Cin k^2) folded into the kernel's weight image."""

    def __init__(self, in_channels, out_channels, kernel_size, style_dim, demodulate=True, padding=None, eps=1e-8):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.padding = kernel_size // 2 if padding is None else padding
        self.demodulate, self.eps = demodulate, eps
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = torch.nn.Parameter(torch.zeros(out_channels))
        self.affine = torch.nn.Linear(style_dim, in_channels)
        torch.nn.init.ones_(self.affine.bias)
        self.scale = 1.0 / (in_channels * kernel_size * kernel_size) ** 0.5

    def style(self, latent):
        """The per-input-channel scale [N, Cin] of this layer for a latent [N, style_dim]."""
        return self.affine(latent)

    def forward(self, input, latent=None, *, style=None, input_modulated=False, next_style=None, activation=None, negative_slope=0.2):
        """`style` (= self.style(latent)) may be passed instead of the latent.  A chain of these layers runs as one launch per
        layer when every layer hands the next one's style to its own call:
            s1, s2 = conv1.style(w), conv2.style(w)
            h = conv1(x, style=s1, next_style=s2, activation="leaky_relu")        # stores s2 * lrelu(...)
            y = conv2(h, style=s2, input_modulated=True, activation="leaky_relu")"""
        if style is None:
            style = self.affine(latent)
        return F.modulated_conv2d(input, self.weight, style, self.bias, 1, self.padding, self.demodulate, self.eps,
                                  _wkey=self._wkey_for(self.weight), weight_scale=self.scale, activation=activation,
                                  negative_slope=negative_slope, input_modulated=input_modulated, next_style=next_style)


class BlurLayer(torch.nn.Module):
    """Depthwise 3x3 binomial blur (reference StyleGAN/model.py:140-150)."""

    def forward(self, input):
        return F.blur3x3(input)


class AvgPool2x2(torch.nn.Module):
    """torch.nn.AvgPool2d(2) on the NHWC pooling kernel (reference BigGAN/model.py:188, StyleGAN/model.py:440)."""

    def forward(self, input):
        return F.avg_pool2x2(input)


class UpsampleNearest2x(torch.nn.Module):
    """torch.nn.Upsample(scale_factor=2, mode="nearest") on the NHWC replication kernel (reference BigGAN/model.py:117,
    StyleGAN/model.py:247)."""

    def forward(self, input):
        return F.upsample_nearest2x(input)


class ReflectionPad2d(torch.nn.ReflectionPad2d):
    """torch.nn.ReflectionPad2d on the NHWC padding kernel (reference Pix2Pix/models.py:162,185,204,210) when all four
    sides are padded alike, the tensor lives on the GPU in fp32 or bf16 and its channel count is a multiple of 8;
    anything else (the three-channel image in front of the first convolution) goes to the torch operator."""

    def forward(self, input):
        p = self.padding
        if (input.is_cuda and input.dim() == 4 and len(set(p)) == 1 and input.shape[1] % 8 == 0 and 0 < p[0] < min(input.shape[2:])
                and input.dtype in (torch.float32, torch.bfloat16)):
            return ops.reflect_pad(input, int(p[0]))
        return super().forward(input)


class InstanceNormAct2d(torch.nn.Module):
    """InstanceNorm2d(affine=False, track_running_stats=False) followed by LeakyReLU(negative_slope), or by nothing
    when negative_slope is None, as ONE kernel with hand-written backward and double-backward kernels.  Holds no
    parameters or buffers, so replacing the reference's norm + activation pair (BasicGANs/models.py:79-80) with it
    leaves state_dicts unchanged."""

    def __init__(self, num_features, eps=1e-5, negative_slope=0.2):
        super().__init__()
        self.num_features, self.eps, self.negative_slope = num_features, eps, negative_slope
        self.activation = None if negative_slope is None else "leaky_relu"

    def forward(self, input):
        if input.shape[1] != self.num_features:
            raise RuntimeError(f"expected {self.num_features} channels, got {input.shape[1]}")
        if self.negative_slope is None:
            return ops.instnorm_act(input, self.eps, 0, 0.0)
        return ops.instnorm_act(input, self.eps, 2, float(self.negative_slope))

    def extra_repr(self):
        return f"{self.num_features}, eps={self.eps}, negative_slope={self.negative_slope}"


class BatchNormAct2d(torch.nn.Module):
    """torch.nn.BatchNorm2d followed by ReLU / LeakyReLU / nothing as fused kernels (statistics, finalize, apply; the
    same three for the backward pass).  Parameters, buffers and their names are BatchNorm2d's (weight, bias,
    running_mean, running_var, num_batches_tracked), so it takes the place of the reference's
    `BatchNorm2d(out_dim), ReLU(True)` pair (BasicGANs/models.py:37) without changing a state_dict; the activation
    module that follows it in the reference holds no state.  First-order gradients only; float32, channels % 4 == 0.

    `sync_stats=True` (or convert_sync_batchnorm(net)): in training mode the statistics are those of the batch held by
    all ranks of `process_group` (one process per GPU, each with a shard), i.e. the single-device result on the whole
    batch; default: per-rank statistics, as under the reference's torch.nn.DataParallel."""

    _ACTS = {None: 0, "relu": 1, "leaky_relu": 2}

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, activation="relu",
                 negative_slope=0.2, sync_stats=False, process_group=None):
        super().__init__()
        self.sync_stats, self.process_group = sync_stats, process_group
        if activation not in self._ACTS:
            raise ValueError(f"activation must be one of {list(self._ACTS)}")
        if momentum is None:
            raise ValueError("cumulative moving average (momentum=None) is not supported")
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.activation, self.negative_slope = activation, negative_slope
        if affine:
            self.weight = torch.nn.Parameter(torch.ones(num_features))
            self.bias = torch.nn.Parameter(torch.zeros(num_features))
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)
        if track_running_stats:
            self.register_buffer("running_mean", torch.zeros(num_features))
            self.register_buffer("running_var", torch.ones(num_features))
            self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        else:
            self.register_buffer("running_mean", None)
            self.register_buffer("running_var", None)
            self.register_buffer("num_batches_tracked", None)

    def forward(self, input):
        if input.shape[1] != self.num_features:
            raise RuntimeError(f"expected {self.num_features} channels, got {input.shape[1]}")
        training = self.training or not self.track_running_stats
        act = self._ACTS[self.activation]
        if not training and torch.is_grad_enabled() and (input.requires_grad or (self.affine and self.weight.requires_grad)):
            # evaluation-mode statistics are constants: plain elementwise math, differentiable to any order
            r = torch.rsqrt(self.running_var + self.eps)
            a = r * self.weight if self.affine else r
            b = (self.bias if self.affine else 0) - self.running_mean * a
            y = input * a[None, :, None, None] + b[None, :, None, None]
            if act == 1:
                return torch.relu(y)
            return torch.nn.functional.leaky_relu(y, self.negative_slope) if act == 2 else y
        if training and self.sync_stats:
            y, mean, var = ops.sync_batchnorm_act(input, self.weight, self.bias, self.eps, act, self.negative_slope, self.process_group)
            if self.track_running_stats:
                with torch.no_grad():
                    ops.batchnorm_update_running_(self.running_mean, self.running_var, self.num_batches_tracked, mean, var, float(self.momentum))
            return y
        if training:      # (the running statistics are updated by the forward pass's own launches)
            return ops.batchnorm_act_train(input, self.weight, self.bias, self.running_mean, self.running_var, self.num_batches_tracked,
                                           self.momentum, self.eps, act, self.negative_slope)
        return ops.batchnorm_act(input, self.weight, self.bias, self.running_mean, self.running_var, False, float(self.eps), act,
                                 float(self.negative_slope))[0]

    def extra_repr(self):
        return (f"{self.num_features}, eps={self.eps}, momentum={self.momentum}, affine={self.affine}, "
                f"track_running_stats={self.track_running_stats}, activation={self.activation}")


def convert_sync_batchnorm(module, process_group=None):
    """Switches every BatchNormAct2d of `module` to statistics over all ranks of `process_group` (the counterpart of
    torch.nn.SyncBatchNorm.convert_sync_batchnorm for this package's fused layer).  Returns `module`."""
    for m in module.modules():
        if isinstance(m, BatchNormAct2d):
            m.sync_stats, m.process_group = True, process_group
        elif isinstance(m, (Conv2d, ConvTranspose2d)):
            m.channel_statistics = False       # (rows of one rank's sums are of no use to statistics over all ranks)
    return module
