"""BigGAN at 128x128 (class-conditional; spectral normalisation, conditional BatchNorm, one self-attention block per
network, projection discriminator) on the gan_amd kernels, assembled from two block tables.

Kept from BigGAN/model.py, because checkpoints and seeded runs depend on it: the state_dict keys and shapes
(`main_sequence.1.conv_2.U_vector`, `main_sequence.0.batch_norm_1.linear.weight`, `bn.running_var`, ...), the order in
which modules with parameters are created and registered (so a seeded construction followed by the reference's
orthogonal_init, BigGAN/train.py:108-112, draws the same numbers), and the arithmetic including its quirks: the spectral
norm sigma is a constant in the backward pass (:27-35), the conditional gain has no "+1" (:68-77), the attention product
[B, positions, C] is reinterpreted as [B, C, H, W] without a transpose (:103), and the projection term multiplies the
class embedding with the [B, 1] score (:233).

What runs where: every convolution (3x3, 1x1, the attention projections) on the gan_amd convolution kernels; conditional
BatchNorm + ReLU on the fused normalisation kernels; 2x2 average pooling and nearest x2 up-scaling on the resampling
kernels; the ReLU between a discriminator block's two convolutions in the first one's epilogue; the power iteration of the
spectral norm and the orthogonal regularisation on library kernels (ops.spectral_sigma_, ops.orthogonal_regularization_);
the spectrally normalised Linear layers, the embeddings and (by default) the attention product on torch ops.  A
discriminator block's 1x1 shortcut runs behind the pooling instead of in front of it (both are linear: same result, a
quarter of the pixels).

Options (off by default): set_activation_dtype(torch.bfloat16) stores activations in bf16 between layers;
set_fused_attention(True) computes softmax(K Q) V with ops.attention, which never materialises the [B, 4096, 4096]
weights.  Phases: hinge loss of BasicGANs/loss_functions.py:79-99, step order and orthogonal regularisation of
BigGAN/train.py:115-127, 187-226.
"""
import contextlib
import os

import torch
import torch.nn.functional as TF

from .. import functional as GF
from .. import nn as gnn
from .. import ops

# channel multipliers (of `ch`) per residual block; "attention" marks where the self-attention block sits
GENERATOR_PLAN = ((16, 16), (16, 8), (8, 4), (4, 2), "attention", (2, 1))                  # BigGAN/model.py:147-154
CRITIC_PLAN = ((None, 1, True), "attention", (1, 2, True), (2, 4, True), (4, 8, True), (8, 16, True), (16, 16, False))   # :212-220

_opt = {"dtype": None, "fused_attention": False, "fused_norm": True, "defer": True,
        "divided_pack": os.environ.get("GC_NO_DIVIDED_PACK") is None}


def set_activation_dtype(dtype):
    """torch.bfloat16: convolutions store bf16 outputs; None or torch.float32: fp32 storage (the reference's)."""
    _opt["dtype"] = None if dtype in (None, torch.float32) else dtype


def set_fused_attention(on):
    _opt["fused_attention"] = bool(on)


def set_deferred_activations(on):
    """Whether the discriminator blocks leave their ReLU derivatives to the consuming data-gradient kernels (default)."""
    _opt["defer"] = bool(on)


def set_divided_pack(on):
    """Whether the convolutions' packed weight images are divided by sigma when they are packed, all layers of a network in
    one launch per forward pass (default), or every forward call forms the tensor weight / sigma and packs it."""
    _opt["divided_pack"] = bool(on)


def set_fused_norm(on):
    """Training-mode (conditional) BatchNorm + ReLU on the fused normalisation kernels (default) or on torch ops."""
    _opt["fused_norm"] = bool(on)


# ---- spectral normalisation --------------------------------------------------------------------------------------------
def _sigma(layer):
    """Largest singular value of layer.weight (as a [out, rest] matrix) by power iteration from the stored left vector
    U_vector [1, out], which is advanced in place.  No gradient flows through it."""
    ahead = layer.__dict__.pop("_sigma_ahead", None)      # computed for the whole network in front of this forward pass
    if ahead is not None and ahead[1] == layer.weight._version:      # (a sigma left over from a pass that skipped this layer,
        return ahead[0]                                               #  or from before the weights moved, is not used)
    layer.__dict__.pop("_divided", None)                  # (its packed images hold another sigma than the one made below)
    if layer.weight.is_cuda and layer.norm == 2 and layer.weight.dtype == torch.float32:
        return ops.spectral_sigma_(layer.weight, layer.U_vector, layer.n_iterations)       # four launches per iteration
    with torch.no_grad():
        w = layer.weight.flatten(1)
        for _ in range(layer.n_iterations):
            v = TF.normalize(layer.U_vector @ w, p=layer.norm)
            u = TF.normalize(v @ w.t(), p=layer.norm)
            s = (u @ w @ v.t()).reshape(())
            layer.U_vector.copy_(u)        # in place: the buffer keeps its storage, which a recorded step relies on
    return s


def _sigmas_ahead(net, need_backward=True):
    """One power iteration per spectrally normalised layer of `net`, all layers in one set of launches; each layer's next
    forward call takes its sigma from here instead of iterating itself (same values: the weights do not move inside a
    forward pass).  Layers that cannot take part (CPU, another norm) keep iterating in their own forward call."""
    layers = net.__dict__.get("_sn_layers")
    if layers is None:
        layers = [m for m in net.modules() if isinstance(m, (Conv2d, Linear)) and m.norm == 2 and m.n_iterations == 1]
        net.__dict__["_sn_layers"] = layers
        net.__dict__["_sn_set"] = None
    if not layers or not layers[0].weight.is_cuda or any(m.weight.dtype != torch.float32 for m in layers):
        return
    st = net.__dict__["_sn_set"]
    if st is None:
        st = net.__dict__["_sn_set"] = ops.SpectralNormSet([m.weight for m in layers], [m.U_vector for m in layers])
    else:
        st.weights, st.us = [m.weight for m in layers], [m.U_vector for m in layers]       # (re-created tensors show in the key)
    sig = st.refresh()
    for i, m in enumerate(layers):
        m.__dict__["_sigma_ahead"] = (sig[i], m.weight._version)
    if _opt["divided_pack"]:
        # the packed images of the convolutions' weights, divided by the new sigmas, in one launch; each convolution is then
        # called with its weight itself (_OverSigma).  Not for the up-sampling convolutions: their image is packed from the
        # folded 4 x 4 filter, a tensor of its own per call.
        idx = [i for i, m in enumerate(layers) if isinstance(m, Conv2d) and not m.__dict__.get("_upsampling")]
        if idx:
            convs = [layers[i] for i in idx]
            epoch = ops.repack_divided([m._wkey_for(m.weight) for m in convs], [m.weight for m in convs], sig, idx,
                                       forward_only=not need_backward)
            for m in convs:
                m.__dict__["_divided"] = epoch


class _OverSigma(torch.autograd.Function):
    """The weight itself, standing for weight / sigma in a convolution whose packed images were divided by sigma when they
    were packed (ops.repack_divided); the gradient that comes back is the one with respect to weight / sigma and is divided
    here (sigma is a constant of the backward pass, BigGAN/model.py:27-35).  The images of a layer are those of its LATEST
    forward pass: a backward pass through an earlier one would have read the wrong sigma and is refused."""

    @staticmethod
    def forward(ctx, weight, sigma, key, epoch):
        ctx.save_for_backward(sigma)
        ctx.key, ctx.epoch = key, epoch
        return weight.view_as(weight)

    @staticmethod
    def backward(ctx, g):
        if ops.divisor_epoch(ctx.key) != ctx.epoch:
            raise RuntimeError("ganconv: backward pass through a spectrally normalised convolution that has run another forward pass "
                               "since (its packed weights follow the latest sigma); run each backward pass before the layer's next "
                               "forward pass, or biggan.set_divided_pack(False)")
        (sigma,) = ctx.saved_tensors
        return g / sigma, None, None, None


class Conv2d(gnn.Conv2d):
    """Convolution with the weight W / sigma(W)."""

    def __init__(self, input_channels, output_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 n_iterations=1, norm=2):
        super().__init__(input_channels, output_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                         groups=groups, bias=bias)
        self.n_iterations, self.norm = n_iterations, norm
        self.register_buffer("U_vector", torch.rand(1, output_channels))

    def forward(self, x, activation=None, out_dtype=None, upsample=False, defer=False, in_slope=None):
        """upsample: the input is replicated x2 first (BigGAN/model.py:117-119), folded into the 3x3 weight.  defer / in_slope:
        the ReLU's derivative is applied by the data-gradient kernel of the one convolution that reads the activated tensor
        (functional.conv2d's defer_activation_grad / input_activation_slope)."""
        out_dtype = out_dtype if out_dtype is not None else _opt["dtype"]
        sigma = _sigma(self)
        epoch = self.__dict__.pop("_divided", None)
        if upsample:
            self.__dict__["_upsampling"] = True
            return GF.upsample2x_conv3x3(x, self.weight / sigma, self.bias, activation=activation, out_dtype=out_dtype)
        key = self._wkey_for(self.weight) if epoch is not None else 0
        if key and ops.divisor_epoch(key) == epoch:
            weight = _OverSigma.apply(self.weight, sigma, key, epoch)
        else:
            weight, key = self.weight / sigma, 0
        return GF.conv2d(x, weight, self.bias, self.stride, self.padding, self.dilation, self.groups, _wkey=key,
                         activation=activation, out_dtype=out_dtype, defer_activation_grad=defer, input_activation_slope=in_slope)


class Linear(torch.nn.Linear):
    def __init__(self, in_features, out_features, bias=True, n_iterations=1, norm=2):
        super().__init__(in_features, out_features, bias=bias)
        self.n_iterations, self.norm = n_iterations, norm
        self.register_buffer("U_vector", torch.rand(1, out_features))

    def forward(self, x):
        return TF.linear(x, self.weight / _sigma(self), self.bias)


# ---- normalisation -----------------------------------------------------------------------------------------------------
def _batchnorm_relu(bn, x, gain, shift):
    """relu(gain * BatchNorm(x) + shift); gain / shift: [C] parameters or per-sample [N, C] tensors."""
    if _opt["fused_norm"] and bn.training and x.shape[1] % 4 == 0:
        track = bn.track_running_stats
        return ops.batchnorm_act_train(x, gain, shift, bn.running_mean if track else None, bn.running_var if track else None,
                                       bn.num_batches_tracked if track else None, bn.momentum, bn.eps, 1, 0.0)
    if gain.dim() == 2:
        gain, shift = gain[:, :, None, None], shift[:, :, None, None]
        return torch.relu(gain.to(x.dtype) * bn(x) + shift.to(x.dtype))
    return torch.relu(bn(x))          # (bn is affine itself here)


class ConditionalNorm(torch.nn.Module):
    """BatchNorm2d(affine=False) whose gain and shift are linear in the conditioning vector (noise chunk + class embedding)."""

    def __init__(self, channels, y_dim):
        super().__init__()
        self.batch_norm = torch.nn.BatchNorm2d(channels, affine=False)
        self.linear = Linear(y_dim, 2 * channels)

    def forward(self, x, y):
        gain, shift = self.linear(y).chunk(2, dim=1)
        return _batchnorm_relu(self.batch_norm, x, gain, shift)


# ---- blocks ------------------------------------------------------------------------------------------------------------
class SelfAttentionBlock(torch.nn.Module):
    def __init__(self, channels, k=8):
        super().__init__()
        self.f_layer = Conv2d(channels, channels // k, 1)
        self.g_layer = Conv2d(channels, channels // k, 1)
        self.h_layer = Conv2d(channels, channels, 1)
        self.gamma = torch.nn.Parameter(torch.zeros(1, 1))

    def forward(self, x):
        b, c, h, w = x.shape
        n = h * w
        rows = self.f_layer(x).reshape(b, -1, n).transpose(1, 2)       # [b, n, c / k]: softmax rows
        cols = self.g_layer(x).reshape(b, -1, n).transpose(1, 2)       # [b, n, c / k]: softmax columns
        vals = self.h_layer(x).reshape(b, -1, n).transpose(1, 2)       # [b, n, c]
        if _opt["fused_attention"] and ops.get_precision() == ops._lib.GC_PREC_BF16 and n % 128 == 0 and c in (64, 128):
            mixed = ops.attention(rows.bfloat16(), cols.bfloat16(), vals.bfloat16())[0].to(x.dtype)
        else:
            mixed = torch.softmax(rows @ cols.transpose(1, 2), dim=-1) @ vals
        if x.is_cuda and n % 64 == 0 and c % 64 == 0 and _opt["fused_attention"] and x.dtype in (torch.float32, torch.bfloat16):
            return ops.attention_residual(x, mixed.contiguous(), self.gamma)       # the reinterpretation + axpy as one tiled launch
        mixed = mixed.reshape(b, -1, h, w)          # [b, n, c] read as [b, c, h, w], as the reference does
        return torch.addcmul(x, mixed, self.gamma.to(mixed.dtype))


class GeneratorBlock(torch.nn.Module):
    """norm-relu-upsample-conv3x3-norm-relu-conv3x3 plus a 1x1 shortcut on the up-scaled input (BigGAN/model.py:100-133)."""

    def __init__(self, cin, cout, y_dim=148):
        super().__init__()
        self.conv_1 = Conv2d(cin, cout, 3, padding=1)
        self.conv_2 = Conv2d(cout, cout, 3, padding=1)
        self.batch_norm_1 = ConditionalNorm(cin, y_dim)
        self.batch_norm_2 = ConditionalNorm(cout, y_dim)
        self.residual_conv = Conv2d(cin, cout, 1)

    def forward(self, x, y):
        # (the up-sampling convolution writes the statistics rows the conditional BatchNorm behind it reads: no sweep of its own)
        with ops.channel_statistics_request() if (self.training and _opt["fused_norm"] and x.is_cuda) else contextlib.nullcontext():
            h = self.conv_1(self.batch_norm_1(x, y), upsample=True)
        h = self.conv_2(self.batch_norm_2(h, y))
        # (a 1x1 convolution commutes with the replication: it runs on the small image)
        return GF.upsample_nearest2x_add(self.residual_conv(x), h)       # h + the up-sampled shortcut, one sweep


class _ReluDerivativeLater(torch.autograd.Function):
    """relu(x) whose backward pass hands the gradient through unchanged: the convolution that reads the result (its only
    reader) is called with in_slope = 0 and applies the ReLU's derivative in its data-gradient kernel."""

    @staticmethod
    def forward(ctx, x):
        return torch.relu(x)

    @staticmethod
    def backward(ctx, g):
        return g


class DiscriminatorBlock(torch.nn.Module):
    """relu-conv3x3-relu-conv3x3(-pool) plus a 1x1 shortcut.  On the GPU neither ReLU has a backward sweep of its own: the
    derivative of the first rides on conv_1's data-gradient kernel, that of the second (conv_1's epilogue) on conv_2's."""

    def __init__(self, cin, cout, downsample=True):
        super().__init__()
        self.conv_1 = Conv2d(cin, cout, 3, padding=1)
        self.conv_2 = Conv2d(cout, cout, 3, padding=1)
        self.residual_conv = Conv2d(cin, cout, 1)
        self.pool = downsample

    def forward(self, x):
        first = x.is_cuda and _opt["defer"] and x.shape[1] % 4 == 0 and x.requires_grad
        second = x.is_cuda and _opt["defer"] and self.conv_1.out_channels % 4 == 0
        a = _ReluDerivativeLater.apply(x) if first else torch.relu(x)
        h = self.conv_1(a, activation="relu", defer=second, in_slope=0.0 if first else None)
        h = self.conv_2(h, in_slope=0.0 if second else None)
        if self.pool:
            return GF.avg_pool2x2_add(h, self.residual_conv(GF.avg_pool2x2(x)))      # pooled h + shortcut, one sweep
        return h + self.residual_conv(x)


class Generator(torch.nn.Module):
    def __init__(self, n_classes=1000, ch=96):
        super().__init__()
        self.linear_y = torch.nn.Embedding(n_classes, 128)
        self.linear_z = Linear(20, 4 * 4 * 16 * ch)
        self.main_sequence = torch.nn.ModuleList(
            SelfAttentionBlock(2 * ch) if spec == "attention" else GeneratorBlock(spec[0] * ch, spec[1] * ch) for spec in GENERATOR_PLAN)
        self.bn = torch.nn.BatchNorm2d(ch)
        self.to_RGB_layer = Conv2d(ch, 3, 1)

    def forward(self, latent_vector, y):
        _sigmas_ahead(self, torch.is_grad_enabled() and (latent_vector.requires_grad or self.to_RGB_layer.weight.requires_grad))
        chunks = latent_vector.split(20, dim=1)            # the first feeds the 4x4 seed, one more per residual block
        klass = self.linear_y(y)
        h = self.linear_z(chunks[0]).reshape(latent_vector.shape[0], -1, 4, 4)
        if _opt["dtype"] is not None:
            h = h.to(_opt["dtype"])
        used = 1
        for block in self.main_sequence:
            if isinstance(block, GeneratorBlock):
                h = block(h, torch.cat((chunks[used], klass), dim=1))
                used += 1
            else:
                h = block(h)
        h = _batchnorm_relu(self.bn, h, self.bn.weight, self.bn.bias)
        return torch.tanh(self.to_RGB_layer(h, out_dtype=torch.float32))


class Discriminator(torch.nn.Module):
    def __init__(self, n_classes=1000, ch=64, image_dim=3):
        super().__init__()
        blocks = []
        for spec in CRITIC_PLAN:
            if spec == "attention":
                blocks.append(SelfAttentionBlock(ch))
            else:
                blocks.append(DiscriminatorBlock(image_dim if spec[0] is None else spec[0] * ch, spec[1] * ch, spec[2]))
        self.main_sequence = torch.nn.ModuleList(blocks)
        self.embedding_layer = torch.nn.Embedding(n_classes, 16 * ch)      # (not spectrally normalised)
        self.projection_layer = Linear(16 * ch, 1)

    def forward(self, x, y):
        _sigmas_ahead(self, torch.is_grad_enabled() and (x.requires_grad or self.projection_layer.weight.requires_grad))
        h = x
        for block in self.main_sequence:
            h = block(h)
        pooled = torch.relu(h).float().sum(dim=(2, 3))
        score = self.projection_layer(pooled)                                # [B, 1]
        return score + (self.embedding_layer(y) * score).sum(dim=1, keepdim=True)


# ---- training phases -----------------------------------------------------------------------------------------------------
def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad_(flag)


def orthogonal_init(model):
    """BigGAN/train.py:108-112: orthogonal weights for every convolution, Linear layer and embedding, in module order."""
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding)):
            torch.nn.init.orthogonal_(m.weight)


@torch.no_grad()
def orthogonal_regularization(model, beta=1e-4):
    """BigGAN/train.py:115-127: grad += beta * 2 (W W^T with its diagonal removed) W for every weight with more than two
    dimensions, W flattened to [out, rest]."""
    params = [p for p in model.parameters() if p.dim() > 2 and p.grad is not None]
    if params and params[0].is_cuda:
        ops.orthogonal_regularization_(params, beta)       # both products on the convolution kernels
        return
    for p in params:
        w = p.flatten(1)
        gram = w @ w.t()
        gram.fill_diagonal_(0.0)
        p.grad.add_((gram @ w).view_as(p), alpha=2.0 * beta)


def discriminator_phase(net_G, net_D, real, real_label, z, fake_label, orth_reg=True, on_last_backward=None):
    """Hinge loss with the generator frozen (BigGAN/train.py:187-199): one backward pass for the real batch, one for the
    generated batch.  Returns the two loss terms, detached."""
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    loss_real = torch.relu(1.0 - net_D(real, real_label)).mean()
    loss_real.backward()
    fake = net_G(z, fake_label).detach()
    loss_fake = torch.relu(1.0 + net_D(fake, fake_label)).mean()
    if on_last_backward is not None:
        on_last_backward()
    loss_fake.backward()
    if orth_reg:
        orthogonal_regularization(net_D)
    return loss_real.detach(), loss_fake.detach()


def generator_phase(net_G, net_D, z, fake_label, orth_reg=True, on_last_backward=None):
    """BigGAN/train.py:206-217: discriminator frozen, loss -mean D(G(z))."""
    requires_grad(net_G, True)
    requires_grad(net_D, False)
    loss = net_D(net_G(z, fake_label), fake_label).mean().neg()
    if on_last_backward is not None:
        on_last_backward()
    loss.backward()
    if orth_reg:
        orthogonal_regularization(net_G)
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    return loss.detach()
