"""CPU restatement of the reference's BigGAN generator / discriminator and hinge phases on stock torch ops.
TEST INFRASTRUCTURE ONLY (tests/, smoke, cpu_baseline); never imported by the product.

Restated from BigGAN/model.py (SpectralNorm :4-35, Linear :38-45, Conv2d :48-58, conditional BatchNorm :61-77,
SelfAttentionBlock :80-107, GeneratorBlock :110-138, Generator :141-178, DiscriminatorBlock :181-205, Discriminator
:208-235), BasicGANs/loss_functions.py:79-99 and BigGAN/train.py:108-127,187-226.  Construction order and names
equal the reference's.  Pinned by tests/golden/biggan_trace.npz (tests/test_oracle_golden.py).
"""
import torch
import torch.nn.functional as F


class _SN:
    """One power iteration per forward on the flattened weight; sigma enters the graph as a constant."""

    def _sn_init(self, n_out):
        self.register_buffer("U_vector", torch.rand(1, n_out))

    def sn_weight(self):
        W = self.weight.view(self.weight.shape[0], -1)
        with torch.no_grad():
            v = F.normalize(torch.matmul(self.U_vector, W), p=2)
            u = F.normalize(torch.matmul(v, W.T), p=2)
            sigma = torch.matmul(torch.matmul(u, W), v.T).squeeze()
            self.U_vector.data = u
        return self.weight / sigma


class SNLinear(_SN, torch.nn.Linear):
    def __init__(self, cin, cout):
        torch.nn.Linear.__init__(self, cin, cout)
        self._sn_init(cout)

    def forward(self, x):
        return F.linear(x, self.sn_weight(), self.bias)


class SNConv2d(_SN, torch.nn.Conv2d):
    def __init__(self, cin, cout, kernel_size, stride=1, padding=0):
        torch.nn.Conv2d.__init__(self, cin, cout, kernel_size=kernel_size, stride=stride, padding=padding)
        self._sn_init(cout)

    def forward(self, x):
        return F.conv2d(x, self.sn_weight(), self.bias, self.stride, self.padding)


class _CBN(torch.nn.Module):
    def __init__(self, c, y_dim=148):
        super().__init__()
        self.batch_norm = torch.nn.BatchNorm2d(c, affine=False)
        self.linear = SNLinear(y_dim, c * 2)

    def forward(self, x, y):
        gamma, beta = self.linear(y).chunk(2, dim=1)
        return gamma[:, :, None, None] * self.batch_norm(x) + beta[:, :, None, None]


class _Attention(torch.nn.Module):
    def __init__(self, c, k=8):
        super().__init__()
        self.f_layer = SNConv2d(c, c // k, 1)
        self.g_layer = SNConv2d(c, c // k, 1)
        self.h_layer = SNConv2d(c, c, 1)
        self.gamma = torch.nn.Parameter(torch.zeros(1, 1))

    def forward(self, x):
        b, c, h, w = x.shape
        key = self.f_layer(x).view(b, -1, h * w).permute(0, 2, 1)
        query = self.g_layer(x).view(b, -1, h * w)
        att = torch.softmax(torch.bmm(key, query), dim=-1)
        value = self.h_layer(x).view(b, -1, h * w)
        return self.gamma * torch.bmm(att, value.permute(0, 2, 1)).view(b, -1, h, w) + x


class _GBlock(torch.nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv_1 = SNConv2d(cin, cout, 3, 1, 1)
        self.conv_2 = SNConv2d(cout, cout, 3, 1, 1)
        self.batch_norm_1 = _CBN(cin)
        self.batch_norm_2 = _CBN(cout)
        self.residual_conv = SNConv2d(cin, cout, 1, 1, 0)

    def forward(self, x, y):
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        out = self.conv_1(up(F.relu(self.batch_norm_1(x, y))))
        out = self.conv_2(F.relu(self.batch_norm_2(out, y)))
        return out + self.residual_conv(up(x))


class Generator(torch.nn.Module):
    def __init__(self, n_classes=1000, ch=96):
        super().__init__()
        self.linear_y = torch.nn.Embedding(n_classes, 128)
        self.linear_z = SNLinear(20, 4 * 4 * ch * 16)
        self.main_sequence = torch.nn.Sequential(_GBlock(16 * ch, 16 * ch), _GBlock(16 * ch, 8 * ch), _GBlock(8 * ch, 4 * ch),
                                                 _GBlock(4 * ch, 2 * ch), _Attention(2 * ch), _GBlock(2 * ch, ch))
        self.bn = torch.nn.BatchNorm2d(ch)
        self.to_RGB_layer = SNConv2d(ch, 3, 1, 1)

    def forward(self, z, y):
        zs = torch.split(z, 20, dim=1)
        emb = self.linear_y(y)
        out = self.linear_z(zs[0]).view(z.shape[0], -1, 4, 4)
        i = 1
        for layer in self.main_sequence:
            if isinstance(layer, _GBlock):
                out = layer(out, torch.cat((zs[i], emb), dim=1))
                i += 1
            else:
                out = layer(out)
        return torch.tanh(self.to_RGB_layer(F.relu(self.bn(out))))


class _DBlock(torch.nn.Module):
    def __init__(self, cin, cout, downsample=True):
        super().__init__()
        self.conv_1 = SNConv2d(cin, cout, 3, 1, 1)
        self.conv_2 = SNConv2d(cout, cout, 3, 1, 1)
        self.down = downsample
        self.residual_conv = SNConv2d(cin, cout, 1, 1)

    def forward(self, x):
        pool = (lambda t: F.avg_pool2d(t, 2)) if self.down else (lambda t: t)
        return pool(self.conv_2(F.relu(self.conv_1(F.relu(x))))) + pool(self.residual_conv(x))


class Discriminator(torch.nn.Module):
    def __init__(self, n_classes=1000, ch=64, image_dim=3):
        super().__init__()
        self.main_sequence = torch.nn.Sequential(_DBlock(image_dim, ch), _Attention(ch), _DBlock(ch, 2 * ch), _DBlock(2 * ch, 4 * ch),
                                                 _DBlock(4 * ch, 8 * ch), _DBlock(8 * ch, 16 * ch), _DBlock(16 * ch, 16 * ch, downsample=False))
        self.embedding_layer = torch.nn.Embedding(n_classes, 16 * ch)
        self.projection_layer = SNLinear(16 * ch, 1)

    def forward(self, x, y):
        h = torch.sum(F.relu(self.main_sequence(x)), dim=[2, 3])
        out = self.projection_layer(h)
        return out + torch.sum(self.embedding_layer(y) * out, dim=1).unsqueeze(1)


def orthogonal_init(model):
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding)):
            torch.nn.init.orthogonal_(m.weight)


def orthogonal_regularization(module, beta=1e-4):
    with torch.no_grad():
        for p in module.parameters():
            if len(p.shape) > 2 and p.grad is not None:
                w = p.view(p.shape[0], -1)
                p.grad.data += beta * (2 * torch.mm(torch.mm(w, w.t()) * (1. - torch.eye(w.shape[0])), w)).view(p.shape)


def _req(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


def discriminator_phase(net_G, net_D, real, real_label, z, fake_label, orth_reg=True):
    _req(net_G, False); _req(net_D, True)
    d_x = F.relu(1.0 - net_D(real, real_label).view(-1)).mean()
    d_x.backward()
    fake = net_G(z, fake_label)
    d_g_z1 = F.relu(1.0 + net_D(fake.detach(), fake_label).view(-1)).mean()
    d_g_z1.backward()
    if orth_reg:
        orthogonal_regularization(net_D)
    return d_x.detach(), d_g_z1.detach()


def generator_phase(net_G, net_D, z, fake_label, orth_reg=True):
    _req(net_G, True); _req(net_D, False)
    g_error = -net_D(net_G(z, fake_label), fake_label).view(-1).mean()
    g_error.backward()
    if orth_reg:
        orthogonal_regularization(net_G)
    _req(net_G, False); _req(net_D, True)
    return g_error.detach()
