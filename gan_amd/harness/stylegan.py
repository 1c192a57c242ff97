"""StyleGAN-v1 (progressive growing, AdaIN) on the gan_amd kernels: the generator and discriminator that
StyleGAN/model.py builds, assembled here from two layer tables.

What is kept from the reference, because checkpoints and seeded runs depend on it:
  * state_dict keys and tensor shapes (`...convolution_2.conv2d.weight_orig`, `...noise_injection_1.scaling_factor_orig`,
    `...adaIN_1.affine_transformation_layer.linear_layer.bias`, `to_RGB_layers.3.conv2d.bias`, the blur buffers, ...), so
    a checkpoint of StyleGAN/train.py:79-93 loads unchanged;
  * the order in which layers are created and therefore the order of the random draws of a seeded initialisation
    (mapping MLP, then per synthesis block: first convolution, second convolution, the two style layers; then the
    to-image convolutions; the discriminator likewise);
  * the arithmetic, including two quirks: both noise injections of a block use the same noise image
    (StyleGAN/model.py:251-257), and with `use_activations` the discriminator's fade-in blends the block output with its
    own LeakyReLU instead of the down-scaled image (StyleGAN/model.py:492-498).

What is different:
  * the equalized-learning-rate constant sqrt(2 / fan_in) (StyleGAN/model.py:33-37) is not multiplied into a copy of
    the weight by a forward pre-hook on every call; convolutions take the stored parameter `weight_orig` plus the
    constant, which the library folds into its cached weight image (`weight_scale`), and the gradient kernels scale by
    the same constant;
  * the noise -> LeakyReLU -> InstanceNorm -> style chain behind every synthesis convolution is one kernel per direction
    (ops.noise_lrelu_adain); the LeakyReLU behind a discriminator convolution runs in that convolution's epilogue;
    the depthwise blur is the stencil kernel (self-adjoint, so no custom backward);
  * randomness drawn inside the reference's forward (noise images, style-mixing crossover) can be passed in, so that a
    step is repeatable.

set_activation_dtype(torch.bfloat16) stores the activations between the convolutions in bf16 (fp32 accumulation;
images, styles, the minibatch statistic, the final projection and the losses stay fp32).
Training phases: StyleGAN/train.py:211-248 with the loss of BasicGANs/loss_functions.py:4-38.
"""
import math
import random

import torch

from .. import functional as GF
from .. import nn as gnn
from .. import ops

LRELU = 0.2

# (input channels, output channels, how the block gets its input): StyleGAN/model.py:304-314
#   "const": learned 4x4 constant; "nearest": nearest x2 -> 3x3 conv -> blur; "transposed": the 3x3 weight averaged into a
#   4x4 stride-2 transposed conv -> blur when the network is built with fused=True, else as "nearest"
SYNTHESIS_PLAN = ((512, 512, "const"), (512, 512, "nearest"), (512, 512, "nearest"), (512, 512, "nearest"), (512, 256, "nearest"),
                  (256, 128, "transposed"), (128, 64, "transposed"), (64, 32, "transposed"), (32, 16, "transposed"))
# (input channels, output channels, may use the 4x4 stride-2 convolution when fused=True): StyleGAN/model.py:456-466; the last
# block (after the minibatch statistic: 513 channels, 3x3 then 4x4 without padding) is appended by the builder
CRITIC_PLAN = ((16, 32, True), (32, 64, True), (64, 128, True), (128, 256, True), (256, 512, False), (512, 512, False),
               (512, 512, False), (512, 512, False))

_store = {"dtype": None, "fused_chain": True, "defer": True}


def set_activation_dtype(dtype):
    _store["dtype"] = None if dtype in (None, torch.float32) else dtype


def set_fused_chain(on):
    """Noise -> LeakyReLU -> InstanceNorm -> style as one kernel (default) or as separate torch / library ops."""
    _store["fused_chain"] = bool(on)


def set_deferred_activations(on):
    """Whether the critic leaves its LeakyReLU derivatives to the consuming kernels (default) or sweeps them itself."""
    _store["defer"] = bool(on)


def _he_constant(weight):
    """sqrt(2 / fan_in) with the reference's fan_in: every dimension of the weight but the first."""
    return math.sqrt(2.0 / (weight[0].numel()))


# ---- parameters stored under `<name>_orig` ---------------------------------------------------------------------------
class _EqConvCore(gnn.Conv2d):
    """Conv2d whose stored parameter is `weight_orig` ~ N(0, 1); the effective weight is weight_orig * sqrt(2 / fan_in)."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        super().__init__(cin, cout, kernel_size, stride=stride, padding=padding)      # (draws torch's default init first)
        w = self.weight.data.normal_(0.0, 1.0)
        self.bias.data.zero_()
        del self._parameters["weight"]
        self.register_parameter("weight_orig", torch.nn.Parameter(w))
        self.scale = _he_constant(w)

    def effective_weight(self):          # what the reference's forward pre-hook would store in `.weight`
        return self.weight_orig * self.scale

    def forward(self, x, out_dtype=None, upsample=False, defer=False, in_slope=None):
        if x.shape[1] > self.in_channels:
            # an input padded with zero channels to a multiple of 8 (the 513-channel layer behind the minibatch statistic): the
            # weight gets zero input channels to match, so that the kernels read aligned 16-byte channel groups
            w = torch.nn.functional.pad(self.weight_orig, (0, 0, 0, 0, 0, x.shape[1] - self.in_channels))
            return GF.conv2d(x, w, self.bias, self.stride, self.padding, activation=self.activation, negative_slope=self.negative_slope,
                             out_dtype=out_dtype, weight_scale=self.scale, defer_activation_grad=defer, input_activation_slope=in_slope)
        if upsample:      # nearest x2 in front of the 3x3 convolution (StyleGAN/model.py:236-238), folded into the weight
            return GF.upsample2x_conv3x3(x, self.weight_orig, self.bias, activation=self.activation, negative_slope=self.negative_slope,
                                         out_dtype=out_dtype, weight_scale=self.scale)
        return GF.conv2d(x, self.weight_orig, self.bias, self.stride, self.padding, _wkey=self._wkey_for(self.weight_orig),
                         activation=self.activation, negative_slope=self.negative_slope, out_dtype=out_dtype, weight_scale=self.scale,
                         defer_activation_grad=defer, input_activation_slope=in_slope)


class EqualizedConv2d(torch.nn.Module):
    """Holder that gives the convolution the reference's key prefix (`conv2d.`)."""

    def __init__(self, cin, cout, kernel_size, stride=1, padding=0, lrelu=False, keep_fp32=False):
        super().__init__()
        self.conv2d = _EqConvCore(cin, cout, kernel_size, stride, padding)
        if lrelu:
            self.conv2d.activation, self.conv2d.negative_slope = "leaky_relu", LRELU
        self.keep_fp32 = keep_fp32

    def forward(self, x, out_dtype=None, upsample=False, defer=False, in_slope=None):
        """defer / in_slope: the LeakyReLU's derivative is applied by the one consumer of this layer's output (its data-gradient
        kernel or the blur's backward sweep) instead of a sweep of its own; in_slope marks this layer as such a consumer."""
        return self.conv2d(x, torch.float32 if self.keep_fp32 else (out_dtype or _store["dtype"]), upsample, defer, in_slope)


class _EqLinearCore(torch.nn.Linear):
    def __init__(self, cin, cout):
        super().__init__(cin, cout)
        w = self.weight.data.normal_(0.0, 1.0)
        self.bias.data.zero_()
        del self._parameters["weight"]
        self.register_parameter("weight_orig", torch.nn.Parameter(w))
        self.scale = _he_constant(w)

    def effective_weight(self):
        return self.weight_orig * self.scale

    def forward(self, x):
        # (x W^T) * c + b: the constant multiplies the small product instead of the weight matrix
        return torch.addmm(self.bias, x, self.weight_orig.t(), alpha=self.scale)


class EqualizedLinearLayer(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_layer = _EqLinearCore(cin, cout)

    def forward(self, x):
        return self.linear_layer(x)


class NoiseScale(torch.nn.Module):
    """Per-channel noise gain, zero at the start, equalized with fan_in = channels (StyleGAN/model.py:179-185, :219-220)."""

    def __init__(self, channels):
        super().__init__()
        self.scaling_factor_orig = torch.nn.Parameter(torch.zeros(1, channels, 1, 1))
        self.scale = math.sqrt(2.0 / channels)

    def gain(self):
        return self.scaling_factor_orig * self.scale


class StyleAffine(torch.nn.Module):
    """The style layer of AdaIN: w -> (gain, bias) per channel, bias initialised to (1, 0) (StyleGAN/model.py:193-200).
    Keeps the reference's child names; the InstanceNorm2d it also held there has no state."""

    def __init__(self, channels, w_dim):
        super().__init__()
        self.affine_transformation_layer = EqualizedLinearLayer(w_dim, 2 * channels)
        with torch.no_grad():
            self.affine_transformation_layer.linear_layer.bias[:channels] = 1.0
        self.channels, self.eps = channels, 1e-5

    def forward(self, w):
        return self.affine_transformation_layer(w).split(self.channels, dim=1)


class Blur(torch.nn.Module):
    """[1 2 1] x [1 2 1] / 16 per channel.  The two buffers exist only because the reference's state_dict has them."""

    def __init__(self, channels):
        super().__init__()
        k = torch.tensor([1.0, 2.0, 1.0])
        k = (k[:, None] * k[None, :] / 16.0).expand(channels, 1, 3, 3).contiguous()
        self.register_buffer("weight", k)
        self.register_buffer("weight_flip", k.clone())      # the kernel is symmetric: its flip is itself

    def forward(self, x, in_slope=None):
        return GF.blur3x3(x, in_slope)


class LearnedConstant(torch.nn.Module):
    def __init__(self, channels, size=4):
        super().__init__()
        self.constant_input = torch.nn.Parameter(torch.randn(1, channels, size, size))

    def forward(self, batch):
        c = self.constant_input if _store["dtype"] is None else self.constant_input.to(_store["dtype"])
        return c.expand(batch, -1, -1, -1)


class UpsampleConvT(torch.nn.Module):
    """Nearest x2 followed by a 3x3 conv, computed as ONE 4x4 stride-2 transposed conv: each 4x4 tap is the mean of the
    (up to four) 3x3 taps that the x2 replication maps onto it (StyleGAN/model.py:380-406).  `weight` is the 3x3
    parameter [Cin, Cout, 3, 3]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cin, cout, 3, 3))
        self.bias = torch.nn.Parameter(torch.zeros(cout))
        self.scale = math.sqrt(2.0 / (cin * 9))

    def forward(self, x):
        if x.is_cuda:       # the pad + four-shift average as one launch (and one for its gradient)
            w4 = ops.upsample_weight_fold(self.weight, 0, False)                                                # [Cin, Cout, 4, 4]
        else:
            w4 = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(self.weight, [1, 1, 1, 1]), 2, stride=1)
        return GF.conv_transpose2d(x, w4, self.bias, stride=2, padding=1, out_dtype=_store["dtype"], weight_scale=self.scale)


def _numbered(*modules):
    """Children named "0", "1", ... (the reference wraps these groups in Sequential; here they are called one by one)."""
    holder = torch.nn.Module()
    for i, m in enumerate(modules):
        holder.add_module(str(i), m)
    return holder


def _children(holder):
    return list(holder._modules.values())


# ---- generator -------------------------------------------------------------------------------------------------------
def _styled(x, noise, gain_module, styles, style_vector):
    """ys * InstanceNorm(lrelu(x + gain * noise)) + yb."""
    ys, yb = styles(style_vector)
    gain = gain_module.gain()
    if _store["fused_chain"] and x.is_cuda and x.shape[1] % 4 == 0:
        return ops.noise_lrelu_adain(x, noise, gain, ys, yb, styles.eps, LRELU)[0]
    h = torch.nn.functional.leaky_relu(x + gain.to(x.dtype) * noise.to(x.dtype), LRELU)
    h = ops.instnorm_act(h, styles.eps, 0, 0.0) if h.is_cuda else torch.nn.functional.instance_norm(h, eps=styles.eps)
    return ys.to(h.dtype)[:, :, None, None] * h + yb.to(h.dtype)[:, :, None, None]


class SynthesisBlock(torch.nn.Module):
    def __init__(self, cin, cout, kind, w_dim):
        super().__init__()
        self.kind = kind
        self.noise_injection_1 = NoiseScale(cout)
        self.noise_injection_2 = NoiseScale(cout)
        if kind == "const":
            self.convolution_1 = LearnedConstant(cin)
            self.convolution_2 = EqualizedConv2d(cin, cout, 3, padding=1)
        else:
            if kind == "transposed":
                self.convolution_1 = _numbered(UpsampleConvT(cin, cout), Blur(cout))
            else:
                self.convolution_1 = _numbered(gnn.UpsampleNearest2x(), EqualizedConv2d(cin, cout, 3, padding=1), Blur(cout))
            self.convolution_2 = EqualizedConv2d(cout, cout, 3, padding=1)
        self.adaIN_1 = StyleAffine(cout, w_dim)
        self.adaIN_2 = StyleAffine(cout, w_dim)

    def forward(self, x, style_vector, noise=None):
        if self.kind == "const":
            h = self.convolution_1(style_vector.shape[0])
        else:
            h = x
            stages = _children(self.convolution_1)
            if self.kind != "transposed":          # (replication, 3x3 convolution, blur): the first two as one launch
                h = stages[1](h, upsample=True)
                stages = stages[2:]
            for stage in stages:
                h = stage(h)
        if noise is None:
            noise = torch.randn(h.shape[0], 1, h.shape[2], h.shape[3], device=h.device)
        h = _styled(h, noise, self.noise_injection_1, self.adaIN_1, style_vector)
        h = self.convolution_2(h)
        return _styled(h, noise, self.noise_injection_2, self.adaIN_2, style_vector)


class SynthesisNetwork(torch.nn.Module):
    def __init__(self, image_dim=3, fused=True, w_dim=512):
        super().__init__()
        self.synthesis_network = torch.nn.ModuleList(
            SynthesisBlock(cin, cout, kind if (fused or kind != "transposed") else "nearest", w_dim) for cin, cout, kind in SYNTHESIS_PLAN)
        self.to_RGB_layers = torch.nn.ModuleList(EqualizedConv2d(cout, image_dim, 1, keep_fp32=True) for _, cout, _ in SYNTHESIS_PLAN)

    def forward(self, w_vectors, pg_step=8, alpha=1, noises=None, crossover_point=None):
        if len(w_vectors) > 2:
            raise ValueError("style mixing takes at most two latent codes")
        if len(w_vectors) == 2:
            switch = random.randint(1, pg_step) if crossover_point is None else crossover_point
        else:
            switch = pg_step + 1        # never
        h, below = None, None           # features at the current / the previous resolution
        for level in range(pg_step + 1):
            below = h
            w = w_vectors[1] if level >= switch else w_vectors[0]
            h = self.synthesis_network[level](below, w, None if noises is None else noises[level])
        image = self.to_RGB_layers[pg_step](h)
        if 0 <= alpha < 1:              # fade the new resolution in over the up-scaled image of the previous one
            previous = self.to_RGB_layers[pg_step - 1](below)
            previous = torch.nn.functional.interpolate(previous, scale_factor=2, mode="nearest")
            image = torch.lerp(previous, image, alpha)
        return image


class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim=512, w_dim=512, n_layers=8, normalize=True):
        super().__init__()
        self.normalize = normalize
        layers = []
        for i in range(n_layers):
            layers += [EqualizedLinearLayer(z_dim if i == 0 else w_dim, w_dim), torch.nn.LeakyReLU(LRELU)]
        self.mapping_network_layers = torch.nn.Sequential(*layers)

    def forward(self, z):
        if self.normalize:          # pixel-wise normalisation of the latent code (eps 1e-8)
            z = z * torch.rsqrt(z.square().mean(dim=1, keepdim=True) + 1e-8)
        return self.mapping_network_layers(z)


class StyleBasedGenerator(torch.nn.Module):
    def __init__(self, z_dim=512, w_dim=512, n_layers=8, normalize=True, image_dim=3, fused=True):
        super().__init__()
        self.mapping_network = MappingNetwork(z_dim, w_dim, n_layers, normalize)
        self.synthesis_network = SynthesisNetwork(image_dim, fused, w_dim)

    def forward(self, noise_vectors, progressive_growing_phase=8, alpha=1, noises=None, crossover_point=None):
        codes = noise_vectors if isinstance(noise_vectors, (tuple, list)) else [noise_vectors]
        return self.synthesis_network([self.mapping_network(z) for z in codes], progressive_growing_phase, alpha, noises,
                                      crossover_point)


# ---- discriminator ---------------------------------------------------------------------------------------------------
class CriticBlock(torch.nn.Module):
    """conv 3x3 + LeakyReLU, then one of: blur + 4x4 stride-2 conv + LeakyReLU ("strided"), blur + 3x3 conv + average pool +
    LeakyReLU ("pooled"), or a k x k conv + LeakyReLU ("plain", the last block).  The LeakyReLU runs in the epilogue of the
    convolution in front of it, except behind the average pool."""

    def __init__(self, cin, cout, mode, k2=3, p2=1):
        super().__init__()
        self.mode = mode
        self.convolution_1 = _numbered(EqualizedConv2d(cin, cout, 3, padding=1, lrelu=True))
        if mode == "strided":
            self.convolution_2 = _numbered(Blur(cout), EqualizedConv2d(cout, cout, 4, stride=2, padding=1, lrelu=True))
        elif mode == "pooled":
            self.convolution_2 = _numbered(Blur(cout), EqualizedConv2d(cout, cout, 3, padding=1), gnn.AvgPool2x2())
        else:
            self.convolution_2 = EqualizedConv2d(cout, cout, k2, padding=p2, lrelu=True)

    def forward(self, x, out_dtype=None, in_slope=None, defer_out=False):
        """in_slope: x is the activated output of the block in front, which left its LeakyReLU's derivative to this block's first
        data-gradient kernel; defer_out: this block leaves its last LeakyReLU's derivative to the next block in the same way.
        Inside the block the blur's backward sweep applies the first LeakyReLU's derivative.  Results are unchanged; the
        backward pass loses one elementwise sweep per LeakyReLU."""
        fuse = _store["fused_chain"] and _store["defer"] and x.is_cuda
        conv1 = _children(self.convolution_1)[0]
        if self.mode == "plain":
            return self.convolution_2(conv1(x, in_slope=in_slope), out_dtype)
        blur, conv = _children(self.convolution_2)[:2]
        h = blur(conv1(x, defer=fuse, in_slope=in_slope), LRELU if fuse else None)
        if self.mode == "pooled":
            return torch.nn.functional.leaky_relu(_children(self.convolution_2)[2](conv(h, out_dtype)), LRELU)
        return conv(h, out_dtype, defer=defer_out)


class Discriminator(torch.nn.Module):
    def __init__(self, fused=True, use_activations=False):
        super().__init__()
        blocks = [CriticBlock(cin, cout, "strided" if (fused and may_stride) else "pooled") for cin, cout, may_stride in CRITIC_PLAN]
        blocks.append(CriticBlock(513, 512, "plain", k2=4, p2=0))
        self.downsampling_blocks = torch.nn.ModuleList(blocks)
        # one 1x1 from-image convolution per resolution; with use_activations the reference interleaves LeakyReLU modules,
        # which shifts the convolutions to the even indices of this list (state_dict keys from_RGB_layers.{0,2,4,...})
        entry = []
        for cin, _, _ in CRITIC_PLAN + ((512, 512, False),):
            entry.append(EqualizedConv2d(3, cin, 1, lrelu=use_activations))
            if use_activations:
                entry.append(torch.nn.Identity())        # the LeakyReLU's slot: it runs in the convolution's epilogue
        self.from_RGB_layers = torch.nn.ModuleList(entry)
        self.final_projection = EqualizedLinearLayer(512, 1)
        self.use_activations = use_activations

    def forward(self, x, progressive_growing_phase=8, alpha=1):
        pg = progressive_growing_phase
        n = len(self.downsampling_blocks)
        first = n - 1 - pg                          # the block that takes the image's resolution
        entry = self.from_RGB_layers[first * (2 if self.use_activations else 1)]
        # The minibatch statistic divides by a spread over the batch that bf16 rounding can collapse to exactly zero: the
        # features it is taken of (the output of the last block but one) are produced and kept in fp32.
        h = entry(x, torch.float32 if pg == 0 else None)
        fuse = _store["fused_chain"] and _store["defer"] and x.is_cuda
        deferred = False                            # h's LeakyReLU derivative is left to the block that reads h
        for level in range(first, n):
            last = level == n - 1
            if last:                                # minibatch standard deviation as one more feature map
                spread = h.std(dim=0, unbiased=False).mean()
                parts = [h, spread.expand(h.shape[0], 1, h.shape[2], h.shape[3])]
                if _store["fused_chain"] and x.is_cuda and (h.shape[1] + 1) % 8:       # zero channels up to a multiple of 8 (matched by zero weights)
                    parts.append(h.new_zeros(h.shape[0], 8 - (h.shape[1] + 1) % 8, h.shape[2], h.shape[3]))
                h = torch.cat(parts, dim=1)
            block = self.downsampling_blocks[level]
            # the next block's first convolution is the only reader of this block's output: not in front of the statistic's
            # concatenation (and its fp32 features), nor where the output is blended with the down-scaled image
            defer_out = (fuse and block.mode == "strided" and level < n - 2 and not (level == first and 0 <= alpha < 1))
            h = block(h, torch.float32 if level == n - 2 else None, in_slope=LRELU if deferred else None, defer_out=defer_out)
            deferred = defer_out
            if level == first and 0 <= alpha < 1 and not last:
                if self.use_activations:
                    # the reference down-scales the image and converts it, then overwrites the result with LeakyReLU(h):
                    # the blend partner is h's own activation (and the converted image is never used)
                    partner = torch.nn.functional.leaky_relu(h, LRELU)
                else:
                    partner = entry(torch.nn.functional.avg_pool2d(x, 2))
                h = torch.lerp(partner, h, alpha)
        return self.final_projection(h.flatten(1).float())


# ---- training phases (StyleGAN/train.py:211-248) -----------------------------------------------------------------------
def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad_(flag)


def critic_phase(net_G, net_D, real, z, epsilon, pg_step, alpha=1, c_lambda=10.0, noises=None, crossover_point=None,
                 on_last_backward=None):
    """Generator frozen; three backward passes accumulate into the discriminator's gradients: real images (with the
    0.001 drift term), generated images, gradient penalty on their mix (double backward).  Returns
    (D(x) term, D(G(z)), penalty, generated images), all detached."""
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    at = dict(progressive_growing_phase=pg_step, alpha=alpha)
    score_real = net_D(real, **at).flatten()
    d_real = score_real.mean() - 0.001 * score_real.square().mean()
    d_real.neg().backward()
    fake = net_G(z, noises=noises, crossover_point=crossover_point, **at).detach()
    d_fake = net_D(fake, **at).mean()
    d_fake.backward()
    t = epsilon.detach().clone().requires_grad_(True)
    mixed = real.detach() * t + fake * (1 - t)
    (slope,) = torch.autograd.grad(net_D(mixed, **at).sum(), mixed, create_graph=True)
    penalty = c_lambda * (slope.flatten(1).norm(2, dim=1) - 1).square().mean()
    if on_last_backward is not None:
        on_last_backward()
    penalty.backward()
    return d_real.detach(), d_fake.detach(), penalty.detach(), fake


def generator_phase(net_G, net_D, z, pg_step, alpha=1, noises=None, crossover_point=None, on_last_backward=None):
    """Discriminator frozen; the generator's loss is -mean D(G(z)).  Returns (loss, generated images), detached."""
    requires_grad(net_G, True)
    requires_grad(net_D, False)
    at = dict(progressive_growing_phase=pg_step, alpha=alpha)
    fake = net_G(z, noises=noises, crossover_point=crossover_point, **at)
    loss = net_D(fake, **at).mean().neg()
    if on_last_backward is not None:
        on_last_backward()
    loss.backward()
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    return loss.detach(), fake.detach()
