"""CPU restatement of the reference's StyleGAN-v1 generator / discriminator and its WGAN-GP phases on stock torch
ops.  TEST INFRASTRUCTURE ONLY (tests/, smoke, cpu_baseline); never imported by the product.

Restated from StyleGAN/model.py (EqualizedLR :13-50, EqualizedLinearLayer :53-62, EqualizedConv2d :65-79, blur
:82-150, NoiseInput :179-185, AdaIN :188-208, SynthesisNetworkBlock :211-268, MappingNetwork :271-293,
SynthesisNetwork :296-355, StyleBasedGenerator :358-376, FusedUpsample :380-406, DownsampleBlock :409-450,
Discriminator :453-514) and StyleGAN/train.py:211-248 + BasicGANs/loss_functions.py:4-38.  Module construction
order and attribute names equal the reference's, so a seeded initialisation reproduces its weights and its
state_dict keys.  The per-block noise images, the style-mixing crossover and epsilon are arguments.
Pinned by tests/golden/stylegan_trace.npz (tests/test_oracle_golden.py).
"""
import math
import random

import torch
import torch.nn.functional as F


def _equalize(module, name="weight"):
    weight = getattr(module, name)
    del module._parameters[name]
    module.register_parameter(name + "_orig", torch.nn.Parameter(weight.data))

    def hook(mod, inputs):
        w = getattr(mod, name + "_orig")
        setattr(mod, name, w * math.sqrt(2 / (w.data.shape[1] * w.data[0][0].numel())))
    module.register_forward_pre_hook(hook)
    return module


class EqLinear(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        lin = torch.nn.Linear(cin, cout)
        lin.weight.data.normal_(mean=0, std=1)
        lin.bias.data.zero_()
        self.linear_layer = _equalize(lin)

    def forward(self, x):
        return self.linear_layer(x)


class EqConv2d(torch.nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()
        conv = torch.nn.Conv2d(*args, **kwargs)
        conv.weight.data.normal_(mean=0, std=1)
        conv.bias.data.zero_()
        self.conv2d = _equalize(conv)

    def forward(self, x):
        return self.conv2d(x)


class Blur(torch.nn.Module):
    def __init__(self, channel):
        super().__init__()
        w = torch.tensor([[1, 2, 1], [2, 4, 2], [1, 2, 1]], dtype=torch.float32)[None, None]
        w = w / w.sum()
        self.register_buffer("weight", w.repeat(channel, 1, 1, 1))
        self.register_buffer("weight_flip", torch.flip(w, [2, 3]).repeat(channel, 1, 1, 1))

    def forward(self, x):   # plain autograd of the depthwise conv equals the reference's hand-written Functions
        return F.conv2d(x, self.weight, padding=1, groups=x.shape[1])


class _Const(torch.nn.Module):
    def __init__(self, c, size=4):
        super().__init__()
        self.constant_input = torch.nn.Parameter(torch.randn(1, c, size, size))

    def forward(self, x):
        return self.constant_input.repeat(x.shape[0], 1, 1, 1)


class _Noise(torch.nn.Module):
    def __init__(self, c):
        super().__init__()
        self.scaling_factor = torch.nn.Parameter(torch.zeros(1, c, 1, 1))

    def forward(self, x, noise):
        return x + self.scaling_factor * noise


class _AdaIN(torch.nn.Module):
    def __init__(self, c, w_dim=512):
        super().__init__()
        self.instance_normalization = torch.nn.InstanceNorm2d(c)
        self.affine_transformation_layer = EqLinear(w_dim, c * 2)
        self.affine_transformation_layer.linear_layer.bias.data[:c] = 1
        self.affine_transformation_layer.linear_layer.bias.data[c:] = 0

    def forward(self, x, style):
        y_s, y_b = torch.chunk(self.affine_transformation_layer(style), 2, dim=1)
        return y_s[:, :, None, None] * self.instance_normalization(x) + y_b[:, :, None, None]


class _FusedUp(torch.nn.Module):
    def __init__(self, cin, cout, k, padding=0):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(cin, cout, k, k))
        self.bias = torch.nn.Parameter(torch.zeros(cout))
        self.multiplier = math.sqrt(2 / (cin * k * k))
        self.pad = padding

    def forward(self, x):
        w = F.pad(self.weight * self.multiplier, [1, 1, 1, 1])
        w = (w[:, :, 1:, 1:] + w[:, :, :-1, 1:] + w[:, :, 1:, :-1] + w[:, :, :-1, :-1]) / 4
        return F.conv_transpose2d(x, w, self.bias, stride=2, padding=self.pad)


class _SynBlock(torch.nn.Module):
    def __init__(self, cin, cout, first=False, fused=False):
        super().__init__()
        self.noise_injection_1 = _equalize(_Noise(cout), "scaling_factor")
        self.noise_injection_2 = _equalize(_Noise(cout), "scaling_factor")
        if first:
            self.convolution_1 = _Const(cin)
            self.convolution_2 = EqConv2d(cin, cout, kernel_size=3, padding=1)
        else:
            if fused:
                self.convolution_1 = torch.nn.Sequential(_FusedUp(cin, cout, 3, padding=1), Blur(cout))
            else:
                self.convolution_1 = torch.nn.Sequential(torch.nn.Upsample(scale_factor=2, mode="nearest"),
                                                         EqConv2d(cin, cout, kernel_size=3, padding=1), Blur(cout))
            self.convolution_2 = EqConv2d(cout, cout, kernel_size=3, padding=1)
        self.adaIN_1 = _AdaIN(cout)
        self.adaIN_2 = _AdaIN(cout)

    def forward(self, x, style, noise=None):
        out = self.convolution_1(x)
        if noise is None:
            noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3])
        out = self.adaIN_1(F.leaky_relu(self.noise_injection_1(out, noise), 0.2), style)
        out = self.convolution_2(out)
        return self.adaIN_2(F.leaky_relu(self.noise_injection_2(out, noise), 0.2), style)


class _Mapping(torch.nn.Module):
    def __init__(self, z_dim=512, w_dim=512, n_layers=8):
        super().__init__()
        layers = []
        for i in range(n_layers):
            layers += [EqLinear(z_dim if i == 0 else w_dim, w_dim), torch.nn.LeakyReLU(0.2)]
        self.mapping_network_layers = torch.nn.Sequential(*layers)

    def forward(self, z):
        z = z / torch.sqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)
        return self.mapping_network_layers(z)


class _Synthesis(torch.nn.Module):
    def __init__(self, fused=True):
        super().__init__()
        self.synthesis_network = torch.nn.Sequential(
            _SynBlock(512, 512, first=True), _SynBlock(512, 512), _SynBlock(512, 512), _SynBlock(512, 512), _SynBlock(512, 256),
            _SynBlock(256, 128, fused=fused), _SynBlock(128, 64, fused=fused), _SynBlock(64, 32, fused=fused), _SynBlock(32, 16, fused=fused))
        self.to_RGB_layers = torch.nn.Sequential(*[EqConv2d(c, 3, kernel_size=1, stride=1, padding=0)
                                                   for c in (512, 512, 512, 512, 256, 128, 64, 32, 16)])

    def forward(self, ws, pg_step=8, alpha=1, noises=None, crossover_point=None):
        last, w = ws[0], ws[0]
        cross = pg_step + 1
        if len(ws) > 1:
            cross = random.randint(1, pg_step) if crossover_point is None else crossover_point
        for index, block in enumerate(self.synthesis_network):
            if index >= cross:
                w = ws[1]
            cur = block(last, w, None if noises is None else noises[index])
            if index == pg_step:
                break
            last = cur
        out = self.to_RGB_layers[index](cur)
        if 0 <= alpha < 1:
            prev = F.interpolate(self.to_RGB_layers[index - 1](last), scale_factor=2, mode="nearest")
            out = (1 - alpha) * prev + alpha * out
        return out


class StyleBasedGenerator(torch.nn.Module):
    def __init__(self, fused=True):
        super().__init__()
        self.mapping_network = _Mapping()
        self.synthesis_network = _Synthesis(fused=fused)

    def forward(self, zs, progressive_growing_phase=8, alpha=1, noises=None, crossover_point=None):
        if not isinstance(zs, (tuple, list)):
            zs = (zs,)
        return self.synthesis_network([self.mapping_network(z) for z in zs], progressive_growing_phase, alpha, noises, crossover_point)


class _DownBlock(torch.nn.Module):
    def __init__(self, cin, cout, k1, p1, k2=None, p2=None, fused=False, downsample=True):
        super().__init__()
        k2 = k1 if k2 is None else k2
        p2 = p1 if p2 is None else p2
        self.convolution_1 = torch.nn.Sequential(EqConv2d(cin, cout, kernel_size=k1, padding=p1))
        if downsample:
            if fused:
                self.convolution_2 = torch.nn.Sequential(Blur(cout), EqConv2d(cout, cout, kernel_size=4, stride=2, padding=p2))
            else:
                self.convolution_2 = torch.nn.Sequential(Blur(cout), EqConv2d(cout, cout, kernel_size=k2, padding=p2), torch.nn.AvgPool2d(2))
        else:
            self.convolution_2 = EqConv2d(cout, cout, kernel_size=k2, padding=p2)

    def forward(self, x):
        return F.leaky_relu(self.convolution_2(F.leaky_relu(self.convolution_1(x), 0.2)), 0.2)


class Discriminator(torch.nn.Module):
    def __init__(self, fused=True, use_activations=False):
        super().__init__()
        self.downsampling_blocks = torch.nn.Sequential(
            _DownBlock(16, 32, 3, 1, fused=fused), _DownBlock(32, 64, 3, 1, fused=fused), _DownBlock(64, 128, 3, 1, fused=fused),
            _DownBlock(128, 256, 3, 1, fused=fused), _DownBlock(256, 512, 3, 1), _DownBlock(512, 512, 3, 1), _DownBlock(512, 512, 3, 1),
            _DownBlock(512, 512, 3, 1), _DownBlock(513, 512, 3, 1, 4, 0, downsample=False))
        layers = []
        for oc in (16, 32, 64, 128, 256, 512, 512, 512, 512):
            layers.append(EqConv2d(3, oc, kernel_size=1))
            if use_activations:
                layers.append(torch.nn.LeakyReLU(0.2))
        self.from_RGB_layers = torch.nn.Sequential(*layers)
        self.final_projection = EqLinear(512, 1)
        self.use_activations = use_activations

    def forward(self, x, progressive_growing_phase=8, alpha=1):
        pg = progressive_growing_phase
        out = x
        for index in range(pg + 1):
            cur = -pg - 1 + index
            if index == 0:
                r = cur * 2 if self.use_activations else cur
                out = self.from_RGB_layers[r](out)
                if self.use_activations:
                    out = self.from_RGB_layers[r + 1](out)
            if index == pg:
                s = out.std(dim=0, unbiased=False).mean().expand(out.shape[0], 1, out.shape[2], out.shape[3])
                out = torch.cat([out, s], dim=1)
            out = self.downsampling_blocks[cur](out)
            if 0 <= alpha < 1 and index == 0 and cur != -1:
                x = F.avg_pool2d(x, 2)
                r = cur * 2 if self.use_activations else cur
                x = self.from_RGB_layers[r](x)
                if self.use_activations:     # reference quirk: the activation is applied to `out`, not `x`
                    x = self.from_RGB_layers[r + 1](out)
                out = (1 - alpha) * x + alpha * out
        return self.final_projection(out.squeeze(2).squeeze(2))


def _req(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


def critic_phase(net_G, net_D, real, z, epsilon, pg_step, alpha=1, c_lambda=10.0, noises=None, crossover_point=None):
    _req(net_G, False); _req(net_D, True)
    kw = dict(progressive_growing_phase=pg_step, alpha=alpha)
    d_real = net_D(real, **kw).view(-1)
    d_real = d_real.mean() - 0.001 * (d_real ** 2).mean()
    (-d_real).backward()
    fake = net_G(z, noises=noises, crossover_point=crossover_point, **kw)
    d_fake = net_D(fake.detach(), **kw).view(-1).mean()
    d_fake.backward()
    eps = epsilon.detach().clone().requires_grad_(True)
    mixed = real.detach() * eps + fake.detach() * (1 - eps)
    gradient = torch.autograd.grad(inputs=mixed, outputs=net_D(mixed, **kw).sum(), create_graph=True)[0]
    penalty = c_lambda * ((gradient.reshape(gradient.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()
    penalty.backward()
    return d_real.detach(), d_fake.detach(), penalty.detach(), fake.detach()


def generator_phase(net_G, net_D, z, pg_step, alpha=1, noises=None, crossover_point=None):
    _req(net_G, True); _req(net_D, False)
    kw = dict(progressive_growing_phase=pg_step, alpha=alpha)
    fake = net_G(z, noises=noises, crossover_point=crossover_point, **kw)
    g_error = -net_D(fake, **kw).view(-1).mean()
    g_error.backward()
    _req(net_G, False); _req(net_D, True)
    return g_error.detach(), fake.detach()
