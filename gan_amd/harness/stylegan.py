"""StyleGAN-v1 (progressive growing, AdaIN) generator and discriminator of the reference on the gan_amd modules.

Graphs, construction order (so that a seeded initialisation reproduces the reference's weights) and parameter
names follow StyleGAN/model.py: EqualizedLR :13-50, EqualizedLinearLayer :53-62, EqualizedConv2d :65-79, BlurLayer
:132-150, NoiseInput :179-185, AdaptiveInstanceNormalization :188-208, SynthesisNetworkBlock :211-268,
MappingNetwork :271-293, SynthesisNetwork :296-355, StyleBasedGenerator :358-376, FusedUpsample :380-406,
DownsampleBlock :409-450, Discriminator :453-514.  The training phases follow StyleGAN/train.py:211-248 with the
Wasserstein-GP loss of BasicGANs/loss_functions.py:4-38.

What runs where: every Conv2d goes through gan_amd.nn.Conv2d (the EqualizedLR pre-hook's scaled, non-leaf weight is
read at call time), FusedUpsample through gan_amd.functional.conv_transpose2d, the depthwise blur through
gan_amd.functional.blur3x3 (self-adjoint, so it needs no custom backward Function), the InstanceNorm inside AdaIN
through the fused instance-norm kernel; the LeakyReLU that follows a discriminator convolution runs in that
convolution's epilogue (same arithmetic as the separate module); the generator's noise injection -> LeakyReLU ->
InstanceNorm -> style chain is one kernel per direction (gan_amd.ops.noise_lrelu_adain).  The mapping MLP, the style
affine layers, minibatch standard deviation and the fade-in blends stay on stock torch ops.

set_activation_dtype(torch.bfloat16) makes every convolution store bf16 outputs (fp32 accumulation; images, styles,
the minibatch statistic, the final projection and the losses stay fp32); the default keeps fp32 storage.

Randomness the reference draws inside forward (per-block noise images, style-mixing crossover) can be passed in so
that a step is repeatable across devices.  Reference quirks kept: the same noise image is used for both injections
of a block; the discriminator's fade-in with use_activations applies the activation to `output` instead of `x`.
"""
import math
import random

import torch

from .. import functional as GF
from .. import nn as gnn
from .. import ops

_act_dtype = None          # None: every layer follows its input (fp32 storage)


def set_activation_dtype(dtype):
    global _act_dtype
    _act_dtype = None if dtype in (None, torch.float32) else dtype


def _like(t, ref):
    return t if t.dtype == ref.dtype else t.to(ref.dtype)


_fused_chain = True


def set_fused_chain(on):
    """The generator's noise -> LeakyReLU -> InstanceNorm -> style chain on the fused kernel (default) or as modules."""
    global _fused_chain
    _fused_chain = bool(on)


class EqualizedLR:
    def __init__(self, name):
        self.name = name

    def compute_weight(self, module):
        weight = getattr(module, self.name + "_orig")
        fan_in = weight.data.shape[1] * weight.data[0][0].numel()
        return weight * math.sqrt(2 / fan_in)

    @staticmethod
    def apply(module, name):
        fn = EqualizedLR(name)
        weight = getattr(module, name)
        del module._parameters[name]
        module.register_parameter(name + "_orig", torch.nn.Parameter(weight.data))
        module.register_forward_pre_hook(fn)

    def __call__(self, module, input):
        setattr(module, self.name, self.compute_weight(module))


def equalized_lr(module, name="weight"):
    EqualizedLR.apply(module, name)
    return module


class EqualizedLinearLayer(torch.nn.Module):
    def __init__(self, channels_in, channels_out):
        super().__init__()
        linear_layer = torch.nn.Linear(channels_in, channels_out)
        linear_layer.weight.data.normal_(mean=0, std=1)
        linear_layer.bias.data.zero_()
        self.linear_layer = equalized_lr(linear_layer)

    def forward(self, x):
        return self.linear_layer(x)


class EqualizedConv2d(torch.nn.Module):
    def __init__(self, *args, out_dtype=None, **kwargs):
        super().__init__()
        conv2d = gnn.Conv2d(*args, **kwargs)
        conv2d.weight.data.normal_(mean=0, std=1)
        conv2d.bias.data.zero_()
        self.conv2d = equalized_lr(conv2d)
        self.out_dtype = out_dtype        # a fixed storage type for this layer's output (image layers: fp32)

    def fuse_leaky_relu(self, negative_slope):
        self.conv2d.activation, self.conv2d.negative_slope = "leaky_relu", negative_slope
        return self

    def forward(self, x, out_dtype=None):
        self.conv2d.out_dtype = out_dtype or self.out_dtype or _act_dtype
        return self.conv2d(x)


class BlurLayer(torch.nn.Module):
    """Keeps the reference's two buffers (state_dict compatibility); the arithmetic is the fused stencil."""

    def __init__(self, channel):
        super().__init__()
        weight = torch.tensor([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=torch.float32).unsqueeze(0).unsqueeze(0)
        weight /= weight.sum()
        self.register_buffer("weight", weight.repeat(channel, 1, 1, 1))
        self.register_buffer("weight_flip", torch.flip(weight, [2, 3]).repeat(channel, 1, 1, 1))

    def forward(self, x):
        return GF.blur3x3(x)


class PixelWiseNormalization(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.epsilon = 1e-8

    def forward(self, x):
        return x / torch.sqrt(torch.mean(x ** 2, dim=1, keepdim=True) + self.epsilon)


class ConstantInput(torch.nn.Module):
    def __init__(self, n_channels, size=4):
        super().__init__()
        self.constant_input = torch.nn.Parameter(torch.randn(1, n_channels, size, size))

    def forward(self, x):
        c = self.constant_input if _act_dtype is None else self.constant_input.to(_act_dtype)
        return c.repeat(x.shape[0], 1, 1, 1)


class NoiseInput(torch.nn.Module):
    def __init__(self, n_channels):
        super().__init__()
        self.scaling_factor = torch.nn.Parameter(torch.zeros(1, n_channels, 1, 1))

    def forward(self, x, noise):
        return torch.addcmul(x, _like(self.scaling_factor, x), _like(noise, x))     # x + scaling_factor * noise, one pass


class AdaptiveInstanceNormalization(torch.nn.Module):
    def __init__(self, input_ch, w_dim=512):
        super().__init__()
        self.instance_normalization = gnn.InstanceNormAct2d(input_ch, negative_slope=None)
        self.affine_transformation_layer = EqualizedLinearLayer(w_dim, input_ch * 2)
        self.affine_transformation_layer.linear_layer.bias.data[:input_ch] = 1
        self.affine_transformation_layer.linear_layer.bias.data[input_ch:] = 0

    def forward(self, x, style):
        styles = _like(self.affine_transformation_layer(style), x)
        y_s, y_b = torch.chunk(styles, 2, dim=1)
        return torch.addcmul(y_b.unsqueeze(2).unsqueeze(3), y_s.unsqueeze(2).unsqueeze(3), self.instance_normalization(x))


class FusedUpsample(torch.nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, padding=0):
        super().__init__()
        weight = torch.randn(in_channel, out_channel, kernel_size, kernel_size)
        bias = torch.zeros(out_channel)
        self.multiplier = math.sqrt(2 / (in_channel * kernel_size * kernel_size))
        self.weight = torch.nn.Parameter(weight)
        self.bias = torch.nn.Parameter(bias)
        self.pad = padding

    def forward(self, input):
        weight = torch.nn.functional.pad(self.weight * self.multiplier, [1, 1, 1, 1])
        weight = (weight[:, :, 1:, 1:] + weight[:, :, :-1, 1:] + weight[:, :, 1:, :-1] + weight[:, :, :-1, :-1]) / 4
        return GF.conv_transpose2d(input, weight, self.bias, stride=2, padding=self.pad, out_dtype=_act_dtype)


class SynthesisNetworkBlock(torch.nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size=3, padding=1, w_dim=512, first=False, fused=False):
        super().__init__()
        self.noise_injection_1 = equalized_lr(NoiseInput(output_channels), name="scaling_factor")
        self.noise_injection_2 = equalized_lr(NoiseInput(output_channels), name="scaling_factor")
        if first:
            self.convolution_1 = ConstantInput(input_channels, size=4)
            self.convolution_2 = EqualizedConv2d(input_channels, output_channels, kernel_size=kernel_size, padding=padding)
        else:
            if fused:
                self.convolution_1 = torch.nn.Sequential(
                    FusedUpsample(input_channels, output_channels, kernel_size=3, padding=1), BlurLayer(output_channels))
            else:
                self.convolution_1 = torch.nn.Sequential(
                    gnn.UpsampleNearest2x(),
                    EqualizedConv2d(input_channels, output_channels, kernel_size=kernel_size, padding=padding),
                    BlurLayer(output_channels))
            self.convolution_2 = EqualizedConv2d(output_channels, output_channels, kernel_size=kernel_size, padding=padding)
        self.adaIN_1 = AdaptiveInstanceNormalization(output_channels, w_dim=w_dim)
        self.adaIN_2 = AdaptiveInstanceNormalization(output_channels, w_dim=w_dim)
        self.leaky_relu_1 = torch.nn.LeakyReLU(0.2)
        self.leaky_relu_2 = torch.nn.LeakyReLU(0.2)

    def forward(self, x, style, noise=None):
        output = self.convolution_1(x)
        if noise is None:
            b, _, w, h = output.shape
            noise = torch.randn(b, 1, w, h, device=output.device)
        output = self._styled(output, noise, style, self.noise_injection_1, self.leaky_relu_1, self.adaIN_1)
        output = self.convolution_2(output)
        output = self._styled(output, noise, style, self.noise_injection_2, self.leaky_relu_2, self.adaIN_2)
        return output

    @staticmethod
    def _styled(x, noise, style, injection, activation, adain):
        """noise injection -> LeakyReLU -> AdaIN: one fused kernel per direction when it applies (GPU tensor, channels a
        multiple of 4, no second-order graph being built), else the three modules as the reference chains them."""
        if _fused_chain and x.is_cuda and x.shape[1] % 4 == 0:
            w = injection.scaling_factor_orig                      # EqualizedLR: the scaled factor the pre-hook would set
            sf = w * math.sqrt(2 / (w.shape[1] * w[0][0].numel()))
            y_s, y_b = torch.chunk(adain.affine_transformation_layer(style), 2, dim=1)
            return ops.noise_lrelu_adain(x, noise, sf, y_s, y_b, float(adain.instance_normalization.eps),
                                         float(activation.negative_slope))[0]
        return adain(activation(injection(x, noise)), style)


class MappingNetwork(torch.nn.Module):
    def __init__(self, z_dim=512, w_dim=512, n_layers=8, normalize=True):
        super().__init__()
        self.normalization_layer = PixelWiseNormalization() if normalize else torch.nn.Identity()
        layers = []
        for i in range(n_layers):
            layers += [EqualizedLinearLayer(z_dim if i == 0 else w_dim, w_dim), torch.nn.LeakyReLU(0.2)]
        self.mapping_network_layers = torch.nn.Sequential(*layers)

    def forward(self, latent_code):
        return self.mapping_network_layers(self.normalization_layer(latent_code))


class SynthesisNetwork(torch.nn.Module):
    def __init__(self, image_dim=3, fused=True):
        super().__init__()
        self.synthesis_network = torch.nn.Sequential(
            SynthesisNetworkBlock(512, 512, first=True), SynthesisNetworkBlock(512, 512), SynthesisNetworkBlock(512, 512),
            SynthesisNetworkBlock(512, 512), SynthesisNetworkBlock(512, 256), SynthesisNetworkBlock(256, 128, fused=fused),
            SynthesisNetworkBlock(128, 64, fused=fused), SynthesisNetworkBlock(64, 32, fused=fused),
            SynthesisNetworkBlock(32, 16, fused=fused))
        self.to_RGB_layers = torch.nn.Sequential(*[EqualizedConv2d(c, image_dim, kernel_size=1, stride=1, padding=0, out_dtype=torch.float32)
                                                   for c in (512, 512, 512, 512, 256, 128, 64, 32, 16)])

    def forward(self, w_vectors, pg_step=8, alpha=1, noises=None, crossover_point=None):
        assert len(w_vectors) < 3
        output_last = w_vectors[0]
        w_vector = w_vectors[0]
        if len(w_vectors) > 1:
            if crossover_point is None:
                crossover_point = random.randint(1, pg_step)
        else:
            crossover_point = pg_step + 1
        for index, block in enumerate(self.synthesis_network):
            if index >= crossover_point:
                w_vector = w_vectors[1]
            output_cur = block(output_last, w_vector, None if noises is None else noises[index])
            if index == pg_step:
                break
            output_last = output_cur
        output = self.to_RGB_layers[index](output_cur)
        if 0 <= alpha < 1:
            image_last = self.to_RGB_layers[index - 1](output_last)
            image_last = torch.nn.functional.interpolate(image_last, scale_factor=2, mode="nearest")
            output = (1 - alpha) * image_last + alpha * output
        return output


class StyleBasedGenerator(torch.nn.Module):
    def __init__(self, z_dim=512, w_dim=512, n_layers=8, normalize=True, image_dim=3, fused=True):
        super().__init__()
        self.mapping_network = MappingNetwork(z_dim=z_dim, w_dim=w_dim, n_layers=n_layers, normalize=normalize)
        self.synthesis_network = SynthesisNetwork(image_dim=image_dim, fused=fused)

    def forward(self, noise_vectors, progressive_growing_phase=8, alpha=1, noises=None, crossover_point=None):
        if not isinstance(noise_vectors, (tuple, list)):
            noise_vectors = (noise_vectors,)
        w_vectors = [self.mapping_network(z) for z in noise_vectors]
        return self.synthesis_network(w_vectors, progressive_growing_phase, alpha, noises, crossover_point)


class DownsampleBlock(torch.nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size_1, padding_1, kernel_size_2=None, padding_2=None,
                 fused=False, downsample=True):
        super().__init__()
        kernel_size_2 = kernel_size_1 if kernel_size_2 is None else kernel_size_2
        padding_2 = padding_1 if padding_2 is None else padding_2
        self.convolution_1 = torch.nn.Sequential(
            EqualizedConv2d(input_channels, output_channels, kernel_size=kernel_size_1, padding=padding_1))
        self.activation_1 = torch.nn.LeakyReLU(0.2)
        if downsample:
            if fused:
                self.convolution_2 = torch.nn.Sequential(
                    BlurLayer(output_channels),
                    EqualizedConv2d(output_channels, output_channels, kernel_size=4, stride=2, padding=padding_2))
            else:
                self.convolution_2 = torch.nn.Sequential(
                    BlurLayer(output_channels),
                    EqualizedConv2d(output_channels, output_channels, kernel_size=kernel_size_2, padding=padding_2),
                    gnn.AvgPool2x2())
        else:
            self.convolution_2 = EqualizedConv2d(output_channels, output_channels, kernel_size=kernel_size_2, padding=padding_2)
        self.activation_2 = torch.nn.LeakyReLU(0.2)
        # LeakyReLU in the epilogue of the convolution it follows (not after the AvgPool2d of the unfused downsample)
        self.convolution_1[0].fuse_leaky_relu(0.2)
        self._fused_2 = not (downsample and not fused)
        if self._fused_2:
            (self.convolution_2[-1] if downsample else self.convolution_2).fuse_leaky_relu(0.2)

    def forward(self, x, out_dtype=None):
        """`out_dtype`: storage type of this block's result (None: the harness's activation type)."""
        output = self.convolution_1(x)
        if isinstance(self.convolution_2, torch.nn.Sequential):
            for layer in self.convolution_2:
                output = layer(output, out_dtype) if isinstance(layer, EqualizedConv2d) else layer(output)
        else:
            output = self.convolution_2(output, out_dtype)
        return output if self._fused_2 else self.activation_2(output)


class Discriminator(torch.nn.Module):
    def __init__(self, fused=True, use_activations=False):
        super().__init__()
        self.downsampling_blocks = torch.nn.Sequential(
            DownsampleBlock(16, 32, 3, 1, fused=fused), DownsampleBlock(32, 64, 3, 1, fused=fused),
            DownsampleBlock(64, 128, 3, 1, fused=fused), DownsampleBlock(128, 256, 3, 1, fused=fused),
            DownsampleBlock(256, 512, 3, 1), DownsampleBlock(512, 512, 3, 1), DownsampleBlock(512, 512, 3, 1),
            DownsampleBlock(512, 512, 3, 1), DownsampleBlock(513, 512, 3, 1, 4, 0, downsample=False))
        layers = []
        for oc in (16, 32, 64, 128, 256, 512, 512, 512, 512):
            layers.append(EqualizedConv2d(3, oc, kernel_size=1))
            if use_activations:
                layers.append(torch.nn.LeakyReLU(0.2))
        self.from_RGB_layers = torch.nn.Sequential(*layers)
        self.final_projection = EqualizedLinearLayer(512, 1)
        self.use_activations = use_activations

    def forward(self, x, progressive_growing_phase=8, alpha=1):
        output = x
        for index in range(progressive_growing_phase + 1):
            cur_idx = -progressive_growing_phase - 1 + index
            cur_layer = self.downsampling_blocks[cur_idx]
            # The minibatch standard deviation divides by a per-position spread over the batch, which bf16 rounding can
            # collapse to exactly zero: the features it is taken of are produced and kept in fp32.
            feeds_std = index == progressive_growing_phase - 1
            if index == 0:
                cur_idx_rgb = cur_idx if not self.use_activations else cur_idx * 2
                output = self.from_RGB_layers[cur_idx_rgb](output, torch.float32 if progressive_growing_phase == 0 else None)
                if self.use_activations:
                    output = self.from_RGB_layers[cur_idx_rgb + 1](output)
            if index == progressive_growing_phase:
                averaged_std = output.std(dim=0, unbiased=False).mean()
                averaged_std = averaged_std.expand(output.shape[0], 1, output.shape[2], output.shape[3])
                output = torch.cat([output, averaged_std], dim=1)
            output = cur_layer(output, torch.float32 if feeds_std else None)
            if 0 <= alpha < 1 and index == 0 and cur_idx != -1:
                x = torch.nn.functional.avg_pool2d(x, 2)
                cur_idx_rgb = cur_idx if not self.use_activations else cur_idx * 2
                x = self.from_RGB_layers[cur_idx_rgb](x)
                if self.use_activations:
                    x = self.from_RGB_layers[cur_idx_rgb + 1](output)
                output = (1 - alpha) * x + alpha * output
        return self.final_projection(output.squeeze(2).squeeze(2).float())


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def critic_phase(net_G, net_D, real, z, epsilon, pg_step, alpha=1, c_lambda=10.0, noises=None, crossover_point=None,
                 on_last_backward=None):
    """StyleGAN/train.py:211-216: generator frozen, three backward passes into the discriminator."""
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    kw = dict(progressive_growing_phase=pg_step, alpha=alpha)
    d_real = net_D(real, **kw).view(-1)
    d_real = d_real.mean() - 0.001 * (d_real ** 2).mean()
    (-d_real).backward()
    fake = net_G(z, noises=noises, crossover_point=crossover_point, **kw)
    d_fake = net_D(fake.detach(), **kw).view(-1).mean()
    d_fake.backward()
    eps = epsilon.detach().clone().requires_grad_(True)
    mixed = real.detach() * eps + fake.detach() * (1 - eps)
    scores = net_D(mixed, **kw)
    gradient = torch.autograd.grad(inputs=mixed, outputs=scores.sum(), create_graph=True)[0]
    penalty = c_lambda * ((gradient.reshape(gradient.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()
    if on_last_backward is not None:
        on_last_backward()
    penalty.backward()
    return d_real.detach(), d_fake.detach(), penalty.detach(), fake.detach()


def generator_phase(net_G, net_D, z, pg_step, alpha=1, noises=None, crossover_point=None, on_last_backward=None):
    """StyleGAN/train.py:228-234: discriminator frozen."""
    requires_grad(net_G, True)
    requires_grad(net_D, False)
    kw = dict(progressive_growing_phase=pg_step, alpha=alpha)
    fake = net_G(z, noises=noises, crossover_point=crossover_point, **kw)
    g_error = -net_D(fake, **kw).view(-1).mean()
    if on_last_backward is not None:
        on_last_backward()
    g_error.backward()
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    return g_error.detach(), fake.detach()
