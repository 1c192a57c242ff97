"""BigGAN (128x128, class-conditional, spectral norm, conditional BatchNorm, self-attention, projection
discriminator) of the reference on the gan_amd convolution modules.

Graphs, construction order and parameter names follow BigGAN/model.py: SpectralNorm :4-35, Linear :38-45, Conv2d
:48-58, BatchNorm (conditional) :61-77, SelfAttentionBlock :80-107, GeneratorBlock :110-138, Generator :141-178,
DiscriminatorBlock :181-205, Discriminator :208-235; the phases follow BasicGANs/loss_functions.py:79-99 (hinge) and
BigGAN/train.py:108-127,187-226 (orthogonal init / regularisation, step order).

Every Conv2d (3x3, 1x1, incl. the attention projections) runs on the gan_amd kernels through
gan_amd.functional.conv2d with the spectrally normalised weight W / sigma, as the reference calls F.conv2d.  The
conditional BatchNorm + ReLU runs on the fused normalisation kernels, pooling / nearest upsampling on the resampling
kernels; the power iteration, the SN Linear layers, the attention product (unless switched to the fused kernel, below)
and the embeddings stay on stock torch ops.  Reference quirks kept: sigma is not differentiated; the conditional
BN gain has no +1; the projection term broadcasts `output` ([B,1]) against the class embedding.

Two storage / arithmetic options, both off by default (set_activation_dtype, set_fused_attention):
  * activations between layers stored in bf16 (convolutions accumulate in fp32, the normalisation statistics, the
    Linear layers, the losses and the images stay fp32);
  * in the bf16 precision mode, softmax(K Q) V of the attention blocks on gan_amd.ops.attention (bf16 operands, fp32
    accumulation), which never materialises the [B, 4096, 4096] weights the reference's bmm + softmax + bmm writes
    and re-reads (17 GB at batch 256).
"""
import torch

from .. import functional as GF
from .. import nn as gnn
from .. import ops

_act_dtype = None          # None: every layer follows its input (fp32 images -> fp32 activations)
_fused_attention = False


def set_activation_dtype(dtype):
    """torch.bfloat16: convolutions store bf16 outputs; None or torch.float32: fp32 storage (the reference's)."""
    global _act_dtype
    _act_dtype = None if dtype in (None, torch.float32) else dtype


def set_fused_attention(on):
    global _fused_attention
    _fused_attention = bool(on)


def _like(t, ref):
    return t if t.dtype == ref.dtype else t.to(ref.dtype)


_fused_norm = True


def set_fused_norm(on):
    """Training-mode (conditional) BatchNorm + ReLU on the fused normalisation kernels (default) or on torch ops."""
    global _fused_norm
    _fused_norm = bool(on)


def _fused_bn(bn, x, weight, bias, activation):
    """bn: a torch.nn.BatchNorm2d in training mode (its running statistics are updated as torch does); weight / bias:
    its own [C] parameters or per-sample [N, C] gains and biases."""
    y, mean, _, var = ops.batchnorm_act(x, weight, bias, bn.running_mean, bn.running_var, True, float(bn.eps),
                                        1 if activation == "relu" else 0, 0.0)
    if bn.track_running_stats:
        with torch.no_grad():
            ops.batchnorm_update_running_(bn.running_mean, bn.running_var, bn.num_batches_tracked, mean, var, float(bn.momentum))
    return y


class SpectralNorm(object):
    def __init__(self, n_outputs, n_iterations=1, norm=2):
        self.n_iterations = n_iterations
        self.norm = norm
        self.register_buffer("U_vector", torch.rand(1, n_outputs))

    def compute_weight(self):
        W_matrix = self.weight.view(self.weight.shape[0], -1)
        for _ in range(self.n_iterations):
            spectral_norm, _ = self._power_iterations(W_matrix)
        return self.weight / spectral_norm

    def _power_iterations(self, W_matrix):
        with torch.no_grad():
            V_vector = torch.nn.functional.normalize(torch.matmul(self.U_vector, W_matrix), p=self.norm)
            U_vector = torch.nn.functional.normalize(torch.matmul(V_vector, W_matrix.T), p=self.norm)
            spectral_norm = torch.matmul(torch.matmul(U_vector, W_matrix), V_vector.T).squeeze()
            self.U_vector.copy_(U_vector)     # written in place (the reference rebinds .data): same values, and the
            #                                   buffer keeps its storage, which a recorded step relies on
            return spectral_norm, U_vector


class Linear(SpectralNorm, torch.nn.Linear):
    def __init__(self, in_features, out_features, bias=True, n_iterations=1, norm=2):
        torch.nn.Linear.__init__(self, in_features, out_features, bias=bias)
        SpectralNorm.__init__(self, out_features, n_iterations=n_iterations, norm=norm)

    def forward(self, x):
        return torch.nn.functional.linear(x, self.compute_weight(), bias=self.bias)


class Conv2d(SpectralNorm, gnn.Conv2d):
    def __init__(self, input_channels, output_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 n_iterations=1, norm=2):
        gnn.Conv2d.__init__(self, input_channels, output_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                            dilation=dilation, groups=groups, bias=bias)
        SpectralNorm.__init__(self, output_channels, n_iterations=n_iterations, norm=norm)

    def forward(self, x, activation=None, out_dtype=None):
        return GF.conv2d(x, self.compute_weight(), bias=self.bias, stride=self.stride, padding=self.padding,
                         dilation=self.dilation, groups=self.groups, activation=activation,
                         out_dtype=out_dtype if out_dtype is not None else _act_dtype)


class BatchNorm(torch.nn.Module):
    def __init__(self, channels, y_dim=148):
        super().__init__()
        self.batch_norm = torch.nn.BatchNorm2d(channels, affine=False)
        self.linear = Linear(y_dim, channels * 2)

    def forward(self, x, y, activation=None):
        """gamma(y) * BatchNorm(x) + beta(y), then the optional ReLU, as one fused normalisation (statistics, finalize,
        apply) in training mode; the unfused torch composition otherwise."""
        gamma, beta = self.linear(y).chunk(2, dim=1)
        bn = self.batch_norm
        if _fused_norm and bn.training and x.shape[1] % 4 == 0:
            return _fused_bn(bn, x, gamma, beta, activation)
        gamma, beta = _like(gamma, x), _like(beta, x)
        out = gamma.unsqueeze(2).unsqueeze(2) * bn(x) + beta.unsqueeze(2).unsqueeze(2)
        return torch.relu(out) if activation == "relu" else out


class SelfAttentionBlock(torch.nn.Module):
    def __init__(self, input_channels, k=8):
        super().__init__()
        self.f_layer = Conv2d(input_channels, input_channels // k, kernel_size=1, stride=1)
        self.g_layer = Conv2d(input_channels, input_channels // k, kernel_size=1, stride=1)
        self.h_layer = Conv2d(input_channels, input_channels, kernel_size=1, stride=1)
        self.gamma = torch.nn.Parameter(torch.zeros(1, 1))
        self.softmax = torch.nn.Softmax(dim=-1)

    def forward(self, x):
        b, c, h, w = x.shape
        key = self.f_layer(x).reshape(b, -1, h * w).permute(0, 2, 1)      # [b, positions, c / k]
        query = self.g_layer(x).reshape(b, -1, h * w)                      # [b, c / k, positions]
        value = self.h_layer(x).reshape(b, -1, h * w)                      # [b, c, positions]
        if _fused_attention and ops.get_precision() == ops._lib.GC_PREC_BF16 and (h * w) % 128 == 0 and c in (64, 128):
            output = self._fused(key, query.permute(0, 2, 1), value.permute(0, 2, 1), x.dtype)
        else:
            attention_weights = self.softmax(torch.bmm(key, query))
            output = torch.bmm(attention_weights, value.permute(0, 2, 1))
        # the reference reinterprets the [b, positions, c] product as [b, c, h, w] without a transpose
        output = output.reshape(b, -1, h, w)
        # gamma * output + x in one pass whose result takes x's (channels-last) layout
        return torch.addcmul(x, output, _like(self.gamma, output))

    @staticmethod
    def _fused(key, query, value, out_dtype):
        """softmax(key query^T) value on the fused kernel; key, query: [b, positions, c / k], value: [b, positions, c].
        With bf16 channels-last storage the three operands are views of the projections' outputs (no copies)."""
        o, _ = ops.attention(key.to(torch.bfloat16), query.to(torch.bfloat16), value.to(torch.bfloat16))
        return o.to(out_dtype)


class GeneratorBlock(torch.nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size=3, stride=1, padding=1, y_dim=148):
        super().__init__()
        self.conv_1 = Conv2d(input_channels, output_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.conv_2 = Conv2d(output_channels, output_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.relu_1 = torch.nn.ReLU()
        self.relu_2 = torch.nn.ReLU()
        self.upsample = gnn.UpsampleNearest2x()
        self.batch_norm_1 = BatchNorm(input_channels, y_dim=y_dim)
        self.batch_norm_2 = BatchNorm(output_channels, y_dim=y_dim)
        self.residual_conv = Conv2d(input_channels, output_channels, kernel_size=1, stride=1, padding=0)

    def forward(self, x, y):
        output = self.conv_1(self.upsample(self.batch_norm_1(x, y, activation="relu")))     # relu_1 / relu_2 inside the norm
        output = self.conv_2(self.batch_norm_2(output, y, activation="relu"))
        return output + self.residual_conv(self.upsample(x))


class Generator(torch.nn.Module):
    def __init__(self, n_classes=1000, ch=96):
        super().__init__()
        self.linear_y = torch.nn.Embedding(n_classes, 128)
        self.linear_z = Linear(20, 4 * 4 * ch * 16)
        self.main_sequence = torch.nn.Sequential(
            GeneratorBlock(16 * ch, 16 * ch), GeneratorBlock(16 * ch, 8 * ch), GeneratorBlock(8 * ch, 4 * ch),
            GeneratorBlock(4 * ch, 2 * ch), SelfAttentionBlock(2 * ch), GeneratorBlock(2 * ch, 1 * ch))
        self.bn = torch.nn.BatchNorm2d(1 * ch)
        self.to_RGB_layer = Conv2d(1 * ch, 3, kernel_size=1, stride=1)
        self.tanh = torch.nn.Tanh()

    def forward(self, latent_vector, y):
        latent_vectors = torch.split(latent_vector, 20, dim=1)
        y_embedding = self.linear_y(y)
        output = self.linear_z(latent_vectors[0]).view(latent_vector.shape[0], -1, 4, 4)
        if _act_dtype is not None:
            output = output.to(_act_dtype)
        index = 1
        for layer in self.main_sequence:
            if isinstance(layer, GeneratorBlock):
                output = layer(output, torch.cat((latent_vectors[index], y_embedding), dim=1))
                index += 1
            else:
                output = layer(output)
        if _fused_norm and self.bn.training and output.shape[1] % 4 == 0:
            output = _fused_bn(self.bn, output, self.bn.weight, self.bn.bias, "relu")
        else:
            output = torch.nn.functional.relu(self.bn(output))
        return self.tanh(self.to_RGB_layer(output, out_dtype=torch.float32))


class DiscriminatorBlock(torch.nn.Module):
    def __init__(self, input_channels, output_channels, kernel_size=3, stride=1, padding=1, downsample=True):
        super().__init__()
        self.conv_1 = Conv2d(input_channels, output_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.conv_2 = Conv2d(output_channels, output_channels, kernel_size=kernel_size, stride=stride, padding=padding)
        self.relu_1 = torch.nn.ReLU()
        self.relu_2 = torch.nn.ReLU()
        self.downsample = gnn.AvgPool2x2() if downsample else torch.nn.Identity()
        self.residual_conv = Conv2d(input_channels, output_channels, kernel_size=1, stride=1)

    def forward(self, x):
        output = self.downsample(self.conv_2(self.conv_1(self.relu_1(x), activation="relu")))    # relu_2 in conv_1's epilogue
        return output + self.residual_conv(self.downsample(x))     # = downsample(residual_conv(x)): both are linear, the
        #                                                            1x1 convolution then runs on a quarter of the pixels


class Discriminator(torch.nn.Module):
    def __init__(self, n_classes=1000, ch=64, image_dim=3):
        super().__init__()
        self.main_sequence = torch.nn.Sequential(
            DiscriminatorBlock(image_dim, 1 * ch), SelfAttentionBlock(1 * ch), DiscriminatorBlock(1 * ch, 2 * ch),
            DiscriminatorBlock(2 * ch, 4 * ch), DiscriminatorBlock(4 * ch, 8 * ch), DiscriminatorBlock(8 * ch, 16 * ch),
            DiscriminatorBlock(16 * ch, 16 * ch, downsample=False))
        self.relu = torch.nn.ReLU()
        self.embedding_layer = torch.nn.Embedding(n_classes, 16 * ch)
        self.projection_layer = Linear(16 * ch, 1)

    def forward(self, x, y):
        h = torch.sum(self.relu(self.main_sequence(x)).float(), dim=[2, 3])
        output = self.projection_layer(h)
        y_embedding = self.embedding_layer(y)
        return output + torch.sum(y_embedding * output, dim=1).unsqueeze(1)


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad = flag


def orthogonal_init(model):
    """BigGAN/train.py:108-112."""
    for module in model.modules():
        if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding)):
            torch.nn.init.orthogonal_(module.weight)


def orthogonal_regularization(module, beta=1e-4):
    """BigGAN/train.py:115-127: adds beta * 2 (W W^T * (1 - I)) W to the gradient of every parameter with > 2 dims."""
    with torch.no_grad():
        for parameter in module.parameters():
            if len(parameter.shape) > 2 and parameter.grad is not None:
                weight = parameter.view(parameter.shape[0], -1)
                eye = torch.eye(weight.shape[0], device=weight.device)
                gradient = 2 * torch.mm(torch.mm(weight, weight.t()) * (1. - eye), weight)
                parameter.grad.data += beta * gradient.view(parameter.shape)


def discriminator_phase(net_G, net_D, real, real_label, z, fake_label, orth_reg=True, on_last_backward=None):
    """Hinge loss, generator frozen (BigGAN/train.py:187-199); two backward passes."""
    relu = torch.nn.functional.relu
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    d_x = relu(1.0 - net_D(real, real_label).view(-1)).mean()
    d_x.backward()
    fake = net_G(z, fake_label)
    d_g_z1 = relu(1.0 + net_D(fake.detach(), fake_label).view(-1)).mean()
    if on_last_backward is not None:
        on_last_backward()
    d_g_z1.backward()
    if orth_reg:
        orthogonal_regularization(net_D)
    return d_x.detach(), d_g_z1.detach()


def generator_phase(net_G, net_D, z, fake_label, orth_reg=True, on_last_backward=None):
    """BigGAN/train.py:206-217: discriminator frozen."""
    requires_grad(net_G, True)
    requires_grad(net_D, False)
    g_error = -net_D(net_G(z, fake_label), fake_label).view(-1).mean()
    if on_last_backward is not None:
        on_last_backward()
    g_error.backward()
    if orth_reg:
        orthogonal_regularization(net_G)
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    return g_error.detach()
