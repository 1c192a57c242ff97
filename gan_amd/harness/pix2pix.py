"""Pix2Pix U-Net generator, 70x70 PatchGAN discriminator and the training step of the reference on the gan_amd
convolution modules.

Graphs and parameter names follow Pix2Pix/models.py:4-147 (Generator, DownSample, UpSample) and :221-243
(Discriminator), :150-218 (GeneratorHD, ResNetBlock); the step follows Pix2Pix/train.py:94-145.  Reference quirks kept: the U-Net is pre-activation
(LeakyReLU -> conv -> BN going down, ReLU -> transposed conv -> BN going up, `UpSample.leaky_relu` is a ReLU), the
discriminator uses ReLU, dropout(0.5) is always on after the first three up blocks, `interpolation` is ignored.
The dropout masks can be passed in so that a step is repeatable across devices.
"""
import torch

from .. import nn as gnn
from .. import ops as gops

# Speed options, read when a network is BUILT (parameter names and state_dicts are the same either way):
#   activation dtype: storage type of the activations between the layers (None = float32; torch.bfloat16 halves the
#                     traffic of every elementwise pass, accumulation and statistics stay fp32);
#   fused norm:       BatchNorm2d (+ ReLU in the discriminator) on the library's norm kernels instead of torch's.
_ACT_DTYPE = None
_FUSED_NORM = False


def set_activation_dtype(dtype):
    global _ACT_DTYPE
    _ACT_DTYPE = dtype


def set_fused_norm(on):
    global _FUSED_NORM
    _FUSED_NORM = bool(on)


def _batch_norm(ch, activation=None):
    return gnn.BatchNormAct2d(ch, activation=activation) if _FUSED_NORM else torch.nn.BatchNorm2d(ch)


class DownSample(torch.nn.Module):
    """LeakyReLU(0.2) -> Conv2d(4, 2, 1) -> BatchNorm2d (Pix2Pix/models.py:4-17).

    `consumer`, fused-norm networks only: the activation every reader of this block's output applies first ("leaky_relu":
    the next down block, whose own LeakyReLU then is not run again; "relu": the first up block).  It moves into this
    block's last kernel (the norm's apply pass, or the convolution's epilogue without a norm), and the block takes its
    input already activated (`input_activated`).  The up block that reads the same output through the skip connection
    applies ReLU to it, and relu(leaky_relu(v)) = relu(v) with the same derivative, so values and gradients are those
    of the reference graph."""

    def __init__(self, in_ch, out_ch, inplace=False, use_bn=True, consumer=None, input_activated=False):
        super().__init__()
        consumer = consumer if _FUSED_NORM else None
        self.input_activated = bool(input_activated and _FUSED_NORM)
        self.downsample = gnn.Conv2d(in_ch, out_ch, kernel_size=4, stride=2, padding=1, channel_statistics=use_bn and _FUSED_NORM,
                                     activation=None if use_bn else consumer, negative_slope=0.2)
        self.relu = torch.nn.LeakyReLU(0.2, inplace=inplace)
        self.batch_norm = _batch_norm(out_ch, consumer) if use_bn else torch.nn.Identity()

    def forward(self, x):
        return self.batch_norm(self.downsample(x if self.input_activated else self.relu(x)))


class UpSample(torch.nn.Module):
    """ReLU -> ConvTranspose2d(4, 2, 1) -> BatchNorm2d (Pix2Pix/models.py:20-37; `leaky_relu` is a ReLU there).

    `fused`, fused-norm networks only: the input arrives with the ReLU applied (by the block before it and the joining
    kernel of the skip connection), and the ReLU of the NEXT up block -- the only reader of this block's output, through
    dropout (a multiplication by 0 or 2) and the concatenation, both of which commute with it -- runs in this block's norm
    kernel."""

    def __init__(self, in_ch, out_ch, interpolation=False, use_bn=True, fused=False):
        super().__init__()
        if interpolation:
            raise RuntimeError("the bilinear-upsample variant is not on the reference's path (Generator never enables it)")
        self.fused = bool(fused and _FUSED_NORM)
        # (the last block, without BatchNorm, feeds tanh: fused into its epilogue, image in float32)
        self.up_sample = gnn.ConvTranspose2d(in_ch, out_ch, kernel_size=4, stride=2, padding=1, activation=None if use_bn else "tanh",
                                             out_dtype=None if use_bn else torch.float32, channel_statistics=use_bn and _FUSED_NORM)
        self.batch_norm = _batch_norm(out_ch, "relu" if self.fused else None) if use_bn else torch.nn.Identity()
        self.leaky_relu = torch.nn.ReLU()

    def forward(self, x):
        return self.batch_norm(self.up_sample(x if self.fused else self.leaky_relu(x)))


class Generator(torch.nn.Module):
    def __init__(self, ch=64, image_dim=3, interpolation=True):
        super().__init__()
        self.interpolation = interpolation   # accepted and ignored, as in the reference
        # fused-norm networks: every activation of the U-Net runs in the kernel that produces its input (see DownSample /
        # UpSample), and a skip connection's cat + ReLU is one launch (ops.act_concat)
        self.fused = f = _FUSED_NORM
        self.downsample_1 = gnn.Conv2d(image_dim, ch, kernel_size=4, stride=2, padding=1, out_dtype=_ACT_DTYPE,
                                       activation="leaky_relu" if f else None, negative_slope=0.2)
        self.downsample_2 = DownSample(ch, ch * 2, consumer="leaky_relu", input_activated=True)
        self.downsample_3 = DownSample(ch * 2, ch * 4, consumer="leaky_relu", input_activated=True)
        self.downsample_4 = DownSample(ch * 4, ch * 8, consumer="leaky_relu", input_activated=True)
        self.downsample_5 = DownSample(ch * 8, ch * 8, consumer="leaky_relu", input_activated=True)
        self.downsample_6 = DownSample(ch * 8, ch * 8, consumer="leaky_relu", input_activated=True)
        self.downsample_7 = DownSample(ch * 8, ch * 8, consumer="leaky_relu", input_activated=True)
        self.downsample_8 = DownSample(ch * 8, ch * 8, use_bn=False, consumer="relu", input_activated=True)
        self.upsample_1 = UpSample(ch * 8, ch * 8, fused=True)
        self.upsample_2 = UpSample(ch * 8 * 2, ch * 8, fused=True)
        self.upsample_3 = UpSample(ch * 8 * 2, ch * 8, fused=True)
        self.upsample_4 = UpSample(ch * 8 * 2, ch * 8, fused=True)
        self.upsample_5 = UpSample(ch * 8 * 2, ch * 4, fused=True)
        self.upsample_6 = UpSample(ch * 4 * 2, ch * 2, fused=True)
        self.upsample_7 = UpSample(ch * 2 * 2, ch, fused=True)
        self.upsample_8 = UpSample(ch * 2, image_dim, use_bn=False, fused=True)

    def forward(self, x, dropout_masks=None):
        """dropout_masks: optional three tensors (already scaled by 1/(1-p)) multiplied onto the outputs of
        upsample_1..3; default draws them like F.dropout(x, 0.5, training=True)."""
        def drop(t, i):
            if dropout_masks is None:
                return torch.nn.functional.dropout(t, 0.5, training=True)
            return t * dropout_masks[i].to(t.dtype)

        def join(out, skip):       # the up block's input: cat, and in the fused form its ReLU as well
            if self.fused and out.is_cuda and out.dtype == skip.dtype and out.dtype in (torch.float32, torch.bfloat16):
                return gops.act_concat(out, skip, 0.0)
            cat = torch.cat((out, skip), dim=1)
            return torch.relu(cat) if self.fused else cat

        x1 = self.downsample_1(x)
        x2 = self.downsample_2(x1)
        x3 = self.downsample_3(x2)
        x4 = self.downsample_4(x3)
        x5 = self.downsample_5(x4)
        x6 = self.downsample_6(x5)
        x7 = self.downsample_7(x6)
        x8 = self.downsample_8(x7)
        out = drop(self.upsample_1(x8), 0)
        out = drop(self.upsample_2(join(out, x7)), 1)
        out = drop(self.upsample_3(join(out, x6)), 2)
        out = self.upsample_4(join(out, x5))
        out = self.upsample_5(join(out, x4))
        out = self.upsample_6(join(out, x3))
        out = self.upsample_7(join(out, x2))
        return self.upsample_8(join(out, x1))      # tanh in the transposed convolution's epilogue


class ResNetBlock(torch.nn.Module):
    """Residual block of the Pix2PixHD generator (Pix2Pix/models.py:198-218): x + [reflection pad 1, 3x3 conv, BatchNorm,
    ReLU, dropout 0.5, reflection pad 1, 3x3 conv, BatchNorm](x).  `main` keeps the reference's module indices (conv at
    1 and 6, BatchNorm at 2 and 7), so its state_dict loads unchanged; BatchNorm + ReLU run as one fused layer."""

    def __init__(self, ch):
        super().__init__()
        self.main = torch.nn.Sequential(
            gnn.ReflectionPad2d(1), gnn.Conv2d(ch, ch, kernel_size=3, padding=0), gnn.BatchNormAct2d(ch, activation="relu"),
            torch.nn.Identity(), torch.nn.Dropout(0.5),
            gnn.ReflectionPad2d(1), gnn.Conv2d(ch, ch, kernel_size=3, padding=0), gnn.BatchNormAct2d(ch, activation=None))

    def forward(self, x, mask=None):
        h = self.main[2](self.main[1](self.main[0](x)))
        h = self.main[4](h) if mask is None else h * mask.to(h.dtype)
        return x + self.main[7](self.main[6](self.main[5](h)))


class GeneratorHD(torch.nn.Module):
    """Global generator of Pix2PixHD (Pix2Pix/models.py:150-195) on the gan_amd layers: reflection-padded 7x7 convolution
    (49 taps), stride-2 3x3 convolutions, residual blocks, ConvTranspose2d(k3, s2, p1, output_padding=1) -- the data
    gradient kernel with an odd filter on the four output parities --, reflection-padded 7x7 convolution with the tanh in
    its epilogue.  Same constructor, parameter names and initialisation order as the reference."""

    def __init__(self, input_dim=3, output_dim=3, ch=64, n_downsampling=3, n_blocks=9):
        super().__init__()
        nn = torch.nn
        self.main = nn.ModuleList()
        self.main.append(nn.Sequential(gnn.ReflectionPad2d(3), gnn.Conv2d(input_dim, ch, kernel_size=7, padding=0, out_dtype=_ACT_DTYPE),
                                       gnn.BatchNormAct2d(ch, activation="relu"), nn.Identity()))
        for i in range(n_downsampling):
            m = 2 ** i
            self.main.append(nn.Sequential(gnn.Conv2d(ch * m, ch * m * 2, kernel_size=3, stride=2, padding=1),
                                           gnn.BatchNormAct2d(ch * m * 2, activation="relu"), nn.Identity()))
        for i in range(n_blocks):
            self.main.append(ResNetBlock(ch * 2 ** n_downsampling))
        for i in range(n_downsampling):
            m = 2 ** (n_downsampling - i)
            self.main.append(nn.Sequential(gnn.ConvTranspose2d(ch * m, ch * m // 2, kernel_size=3, stride=2, padding=1, output_padding=1),
                                           gnn.BatchNormAct2d(ch * m // 2, activation="relu"), nn.Identity()))
        self.main.append(nn.Sequential(gnn.ReflectionPad2d(3), gnn.Conv2d(ch, output_dim, kernel_size=7, padding=0, activation="tanh", out_dtype=torch.float32),
                                       nn.Identity()))

    def forward(self, x, dropout_masks=None):
        """dropout_masks: optional one tensor per residual block (already scaled by 1 / (1 - p))."""
        out, bi = x, 0
        for layer in self.main:
            if isinstance(layer, ResNetBlock):
                out = layer(out, None if dropout_masks is None else dropout_masks[bi])
                bi += 1
            else:
                out = layer(out)
        return out


class Discriminator(torch.nn.Module):
    def __init__(self, image_dim=6, ch=64):
        super().__init__()
        self.fused = _FUSED_NORM
        self.conv_1 = gnn.Conv2d(image_dim, ch, 4, 2, 1, activation="relu", out_dtype=_ACT_DTYPE)      # ReLU in the epilogue
        self.conv_2 = gnn.Conv2d(ch, ch * 2, 4, 2, 1, channel_statistics=_FUSED_NORM)
        self.batch_norm_1 = _batch_norm(ch * 2, "relu")
        self.conv_3 = gnn.Conv2d(ch * 2, ch * 4, 4, 2, 1, channel_statistics=_FUSED_NORM)
        self.batch_norm_2 = _batch_norm(ch * 4, "relu")
        self.conv_4 = gnn.Conv2d(ch * 4, ch * 8, 4, 1, 1, channel_statistics=_FUSED_NORM)
        self.batch_norm_3 = _batch_norm(ch * 8, "relu")
        self.conv_5 = gnn.Conv2d(ch * 8, 1, 4, 1, 1, out_dtype=torch.float32)

    def forward(self, x, y, return_features=False):
        relu = (lambda t: t) if self.fused else torch.nn.functional.relu       # (fused: inside the norm kernel)
        o1 = self.conv_1(torch.cat((x, y), dim=1))
        o2 = relu(self.batch_norm_1(self.conv_2(o1)))
        o3 = relu(self.batch_norm_2(self.conv_3(o2)))
        o4 = relu(self.batch_norm_3(self.conv_4(o3)))
        o5 = self.conv_5(o4)
        return torch.sigmoid(o5), ([o1, o2, o3, o4, o5] if return_features else None)


def requires_grad(model, flag=True):
    """Pix2Pix/utils.py:43-45."""
    for p in model.parameters():
        p.requires_grad = flag


def discriminator_phase(net_G, net_D, real, label, dropout_masks=None, on_last_backward=None):
    """Pix2Pix/train.py:94-109 up to (not including) optimizer_D.step().  Returns the loss tensor."""
    bce = torch.nn.functional.binary_cross_entropy
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    d_x, _ = net_D(real, label)
    d_x = bce(d_x, torch.ones_like(d_x))
    fake = net_G(label, dropout_masks)
    d_g_z1, _ = net_D(fake.detach(), label)
    d_g_z1 = bce(d_g_z1, torch.zeros_like(d_g_z1))
    loss = (d_x + d_g_z1) * 0.5
    if on_last_backward is not None:
        on_last_backward()
    loss.backward()
    return loss.detach()


def generator_phase(net_G, net_D, real, label, lmbd=100.0, dropout_masks=None, on_last_backward=None):
    """Pix2Pix/train.py:118-139 up to (not including) optimizer_G.step(); BCE + lambda * L1."""
    bce = torch.nn.functional.binary_cross_entropy
    requires_grad(net_G, True)
    requires_grad(net_D, False)
    fake = net_G(label, dropout_masks)
    d_g_z2, _ = net_D(fake, label)
    g_error = bce(d_g_z2, torch.ones_like(d_g_z2)) + lmbd * torch.nn.functional.l1_loss(fake, real)
    if on_last_backward is not None:
        on_last_backward()
    g_error.backward()
    requires_grad(net_G, False)
    requires_grad(net_D, True)
    return g_error.detach(), fake.detach()
