"""CPU restatement of the reference's BasicGANs model graphs and training-step functions on stock torch ops.

TEST INFRASTRUCTURE ONLY.  Nothing the product ships may import this module; only tests/, the smoke check in
__graft_entry__.py and the cpu_baseline leg of bench.py use it, and only as the checker / the CPU timing
baseline.  The reference's own Python files cannot travel to the GPU box, so this file restates them; it is
pinned against golden vectors captured from the real reference files (tests/golden/make_golden.py,
checked by tests/test_oracle_golden.py).

Restated from (paths relative to the reference checkout):
  BasicGANs/models.py:6-41      Generator            -> Generator
  BasicGANs/models.py:44-86     Discriminator        -> Discriminator
  BasicGANs/utils.py:31-36      weights_init         -> weights_init
  BasicGANs/loss_functions.py:4-38   get_wasserstein_loss -> wasserstein_step
  BasicGANs/loss_functions.py:41-76  get_minimax_loss     -> minimax_step
  BasicGANs/DCGAN.py:165-181    one DCGAN G+D step   -> dcgan_train_step
  BasicGANs/train.py:95-119     one WGAN-GP cycle    -> wgangp_train_cycle
  Pix2Pix/models.py:4-147       Generator, DownSample, UpSample -> P2PGenerator, P2PDownSample, P2PUpSample
  Pix2Pix/models.py:221-243     Discriminator        -> P2PDiscriminator
  Pix2Pix/train.py:94-145       one Pix2Pix step     -> pix2pix_discriminator_phase, pix2pix_generator_phase

The convolution layers are taken from a module namespace `nn` (default torch.nn) so that a test can build
the very same graph on top of another Conv2d / ConvTranspose2d implementation and compare.
"""
import math

import torch


def _chain(*mods):
    return torch.nn.Sequential(*mods)


class Generator(torch.nn.Module):
    """z [B, z_dim, 1, 1] -> image [B, image_dim, R, R]: ConvT k4 (s1 p0 first, then s2 p1), BN + ReLU, Tanh last."""

    def __init__(self, z_dim, hidden_dim, image_dim=3, resolution=128, nn=torch.nn):
        super().__init__()
        n_layers = int(math.log2(resolution)) - 1
        width = hidden_dim * 8
        blocks = []
        for i in range(n_layers):
            first, last = i == 0, i == n_layers - 1
            cin = z_dim if first else int(width / 2 ** (i - 1))
            cout = image_dim if last else int(width / 2 ** i)
            ops = [nn.ConvTranspose2d(cin, cout, 4, 1 if first else 2, 0 if first else 1, bias=False)]
            ops += [torch.nn.Tanh()] if last else [torch.nn.BatchNorm2d(cout), torch.nn.ReLU(True)]
            blocks.append(_chain(*ops))
        self.main = _chain(*blocks)

    def forward(self, noise_vector):
        return self.main(noise_vector)


class Discriminator(torch.nn.Module):
    """image -> score [B,1,1,1]: Conv k4 s2 p1 stack (BN or InstanceNorm for the critic, LeakyReLU 0.2),
    final Conv k4 s1 p0, Sigmoid unless is_wgan."""

    def __init__(self, image_dim, hidden_dim, resolution=128, is_wgan=False, nn=torch.nn):
        super().__init__()
        n_layers = int(math.log2(resolution)) - 1
        blocks = []
        for i in range(n_layers):
            first, last = i == 0, i == n_layers - 1
            cin = image_dim if first else int(hidden_dim * 2 ** (i - 1))
            cout = 1 if last else int(hidden_dim * 2 ** i)
            ops = [nn.Conv2d(cin, cout, 4, 1 if last else 2, 0 if last else 1, bias=False)]
            if not first and not last:
                norm = torch.nn.InstanceNorm2d if is_wgan else torch.nn.BatchNorm2d
                ops += [norm(cout), torch.nn.LeakyReLU(0.2, inplace=True)]
            elif last and not is_wgan:
                ops += [torch.nn.Sigmoid()]
            elif first:
                ops += [torch.nn.LeakyReLU(0.2, inplace=True)]
            blocks.append(_chain(*ops))
        self.main = _chain(*blocks)

    def forward(self, x):
        return self.main(x)


def weights_init(m):
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
    if isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
        torch.nn.init.constant_(m.bias, 0)


def wasserstein_step(net_G, net_D, batch=None, noise_vector=None, c_lambda=10, discriminator=True, epsilon=None):
    """One critic or generator phase of WGAN-GP including .backward().  `epsilon` ([B,1,1,1], values in
    [0,1)) replaces the torch.rand draw of the reference so that runs are repeatable across devices."""
    if discriminator:
        d_real = net_D(batch).view(-1)
        d_real = d_real.mean() - 0.001 * (d_real ** 2).mean()
        (-d_real).backward()

        fake = net_G(noise_vector)
        d_fake = net_D(fake.detach()).view(-1).mean()
        d_fake.backward()

        if epsilon is None:
            epsilon = torch.rand(fake.shape[0], 1, 1, 1, device=fake.device)
        epsilon = epsilon.detach().clone().requires_grad_(True)
        mixed = batch.detach() * epsilon + fake.detach() * (1 - epsilon)
        scores = net_D(mixed)
        gradient = torch.autograd.grad(inputs=mixed, outputs=scores.sum(), create_graph=True)[0]
        penalty = ((gradient.view(gradient.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()
        penalty = c_lambda * penalty
        penalty.backward(retain_graph=True)
        return d_real, d_fake.item(), penalty
    fake = net_G(noise_vector)
    d_fake = net_D(fake).view(-1)
    g_error = -d_fake.mean()
    g_error.backward()
    return g_error, d_fake.mean().item()


def minimax_step(net_G, net_D, criterion, label, batch, noise_batch=None, fake_label=None, discriminator=True):
    """One D or G phase of the DCGAN minimax (BCE) objective including .backward()."""
    if discriminator:
        output = net_D(batch).view(-1)
        err_real = criterion(output, label)
        err_real.backward()
        d_x = output.mean().item()
        fake = net_G(noise_batch)
        label.fill_(fake_label)
        output = net_D(fake.detach()).view(-1)
        err_fake = criterion(output, label)
        err_fake.backward()
        return err_real + err_fake, output.mean().item(), d_x, fake
    output = net_D(batch).view(-1)
    g_error = criterion(output, label)
    g_error.backward()
    return g_error, output.mean().item()


def dcgan_train_step(net_G, net_D, opt_G, opt_D, real, z):
    """One DCGAN G+D step; the G phase re-uses the fake batch (and its graph) of the D phase."""
    criterion = torch.nn.BCELoss()
    net_D.zero_grad()
    label = torch.full((real.shape[0],), 1.0, dtype=torch.float, device=real.device)
    d_error, d_g_z1, d_x, fake = minimax_step(net_G, net_D, criterion, label, real, z, 0.0)
    if opt_D is not None:
        opt_D.step()
    net_G.zero_grad()
    label.fill_(1.0)
    g_error, d_g_z2 = minimax_step(net_G, net_D, criterion, label, fake, discriminator=False)
    if opt_G is not None:
        opt_G.step()
    return {"D_error": d_error.item(), "D_G_z1": d_g_z1, "D_x": d_x, "G_error": g_error.item(), "D_G_z2": d_g_z2}


def wgangp_train_cycle(net_G, net_D, opt_G, opt_D, real, zs, eps, c_lambda=10, disc_repeats=5):
    """disc_repeats critic iterations (each with its own z and epsilon) followed by one generator iteration."""
    critic = []
    for it in range(disc_repeats):
        opt_D.zero_grad()
        d_x, d_g_z1, gp = wasserstein_step(net_G, net_D, real, zs[it], c_lambda=c_lambda, epsilon=eps[it])
        opt_D.step()
        critic.append({"D_x": d_x.item(), "D_G_z1": d_g_z1, "gp": gp.item()})
    opt_G.zero_grad()
    g_error, d_g_z2 = wasserstein_step(net_G, net_D, noise_vector=zs[disc_repeats], discriminator=False, c_lambda=c_lambda)
    opt_G.step()
    return {"critic": critic, "G_error": g_error.item(), "D_G_z2": d_g_z2}


def synth_inputs(gen, B, nz, res, n_z):
    """Synthetic real batch U(-1,1) and n_z latent batches N(0,1) from a dedicated CPU generator."""
    real = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
    zs = [torch.randn(B, nz, 1, 1, generator=gen) for _ in range(n_z)]
    return real, zs


# ---- Pix2Pix ------------------------------------------------------------------------------------------------------
class P2PDownSample(torch.nn.Module):
    def __init__(self, in_ch, out_ch, inplace=False, use_bn=True):
        super().__init__()
        self.downsample = torch.nn.Conv2d(in_ch, out_ch, kernel_size=4, stride=2, padding=1)
        self.relu = torch.nn.LeakyReLU(0.2, inplace=inplace)
        self.batch_norm = torch.nn.BatchNorm2d(out_ch) if use_bn else torch.nn.Identity()

    def forward(self, x):
        return self.batch_norm(self.downsample(self.relu(x)))


class P2PUpSample(torch.nn.Module):
    def __init__(self, in_ch, out_ch, use_bn=True):
        super().__init__()
        self.up_sample = torch.nn.ConvTranspose2d(in_ch, out_ch, kernel_size=4, stride=2, padding=1)
        self.batch_norm = torch.nn.BatchNorm2d(out_ch) if use_bn else torch.nn.Identity()
        self.leaky_relu = torch.nn.ReLU()

    def forward(self, x):
        return self.batch_norm(self.up_sample(self.leaky_relu(x)))


class P2PGenerator(torch.nn.Module):
    """U-Net: 8 stride-2 convs down (256 -> 1), 8 stride-2 transposed convs up with skip concatenation, dropout
    0.5 (always on) after the first three up blocks, tanh.  Construction order = the reference's, so that a seeded
    default initialisation reproduces the reference's initial weights."""

    def __init__(self, ch=64, image_dim=3):
        super().__init__()
        self.downsample_1 = torch.nn.Conv2d(image_dim, ch, kernel_size=4, stride=2, padding=1)
        self.downsample_2 = P2PDownSample(ch, ch * 2)
        self.downsample_3 = P2PDownSample(ch * 2, ch * 4)
        self.downsample_4 = P2PDownSample(ch * 4, ch * 8)
        self.downsample_5 = P2PDownSample(ch * 8, ch * 8)
        self.downsample_6 = P2PDownSample(ch * 8, ch * 8)
        self.downsample_7 = P2PDownSample(ch * 8, ch * 8)
        self.downsample_8 = P2PDownSample(ch * 8, ch * 8, use_bn=False)
        self.upsample_1 = P2PUpSample(ch * 8, ch * 8)
        self.upsample_2 = P2PUpSample(ch * 8 * 2, ch * 8)
        self.upsample_3 = P2PUpSample(ch * 8 * 2, ch * 8)
        self.upsample_4 = P2PUpSample(ch * 8 * 2, ch * 8)
        self.upsample_5 = P2PUpSample(ch * 8 * 2, ch * 4)
        self.upsample_6 = P2PUpSample(ch * 4 * 2, ch * 2)
        self.upsample_7 = P2PUpSample(ch * 2 * 2, ch)
        self.upsample_8 = P2PUpSample(ch * 2, image_dim, use_bn=False)

    def forward(self, x, dropout_masks=None):
        def drop(t, i):
            return torch.nn.functional.dropout(t, 0.5, training=True) if dropout_masks is None else t * dropout_masks[i]
        x1 = self.downsample_1(x); x2 = self.downsample_2(x1); x3 = self.downsample_3(x2); x4 = self.downsample_4(x3)
        x5 = self.downsample_5(x4); x6 = self.downsample_6(x5); x7 = self.downsample_7(x6); x8 = self.downsample_8(x7)
        out = drop(self.upsample_1(x8), 0)
        out = drop(self.upsample_2(torch.cat((out, x7), dim=1)), 1)
        out = drop(self.upsample_3(torch.cat((out, x6), dim=1)), 2)
        out = self.upsample_4(torch.cat((out, x5), dim=1))
        out = self.upsample_5(torch.cat((out, x4), dim=1))
        out = self.upsample_6(torch.cat((out, x3), dim=1))
        out = self.upsample_7(torch.cat((out, x2), dim=1))
        out = self.upsample_8(torch.cat((out, x1), dim=1))
        return torch.tanh(out)


class P2PDiscriminator(torch.nn.Module):
    """70x70 PatchGAN on cat(image, label): k4 convs s2,s2,s2,s1,s1 with BatchNorm on 2-4, ReLU, sigmoid."""

    def __init__(self, image_dim=6, ch=64):
        super().__init__()
        self.conv_1 = torch.nn.Conv2d(image_dim, ch, 4, 2, 1)
        self.conv_2 = torch.nn.Conv2d(ch, ch * 2, 4, 2, 1)
        self.batch_norm_1 = torch.nn.BatchNorm2d(ch * 2)
        self.conv_3 = torch.nn.Conv2d(ch * 2, ch * 4, 4, 2, 1)
        self.batch_norm_2 = torch.nn.BatchNorm2d(ch * 4)
        self.conv_4 = torch.nn.Conv2d(ch * 4, ch * 8, 4, 1, 1)
        self.batch_norm_3 = torch.nn.BatchNorm2d(ch * 8)
        self.conv_5 = torch.nn.Conv2d(ch * 8, 1, 4, 1, 1)

    def forward(self, x, y):
        relu = torch.nn.functional.relu
        o = relu(self.conv_1(torch.cat((x, y), dim=1)))
        o = relu(self.batch_norm_1(self.conv_2(o)))
        o = relu(self.batch_norm_2(self.conv_3(o)))
        o = relu(self.batch_norm_3(self.conv_4(o)))
        return torch.sigmoid(self.conv_5(o))


class P2PResNetBlock(torch.nn.Module):
    """x + [reflection pad 1, 3x3 conv, BatchNorm, ReLU, dropout 0.5, reflection pad 1, 3x3 conv, BatchNorm](x)
    (Pix2Pix/models.py:198-218).  Module indices inside `main` are the reference's, so state_dict keys agree."""

    def __init__(self, ch):
        super().__init__()
        self.main = torch.nn.Sequential(
            torch.nn.ReflectionPad2d(1), torch.nn.Conv2d(ch, ch, kernel_size=3, padding=0), torch.nn.BatchNorm2d(ch),
            torch.nn.ReLU(), torch.nn.Identity(),
            torch.nn.ReflectionPad2d(1), torch.nn.Conv2d(ch, ch, kernel_size=3, padding=0), torch.nn.BatchNorm2d(ch))

    def forward(self, x, mask=None):
        h = self.main[3](self.main[2](self.main[1](self.main[0](x))))
        h = torch.nn.functional.dropout(h, 0.5, training=self.training) if mask is None else h * mask
        return x + self.main[7](self.main[6](self.main[5](h)))


class P2PGeneratorHD(torch.nn.Module):
    """Global generator of Pix2PixHD as the reference builds it (Pix2Pix/models.py:150-195): reflection-padded 7x7 conv,
    n_downsampling stride-2 3x3 convs, n_blocks residual blocks, as many ConvTranspose2d(k3, s2, p1, output_padding 1),
    reflection-padded 7x7 conv to the image, tanh; BatchNorm + ReLU after every conv but the last."""

    def __init__(self, input_dim=3, output_dim=3, ch=64, n_downsampling=3, n_blocks=9):
        super().__init__()
        nn = torch.nn
        self.main = nn.ModuleList()
        self.main.append(nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(input_dim, ch, kernel_size=7, padding=0), nn.BatchNorm2d(ch), nn.ReLU()))
        for i in range(n_downsampling):
            m = 2 ** i
            self.main.append(nn.Sequential(nn.Conv2d(ch * m, ch * m * 2, kernel_size=3, stride=2, padding=1), nn.BatchNorm2d(ch * m * 2), nn.ReLU()))
        for i in range(n_blocks):
            self.main.append(P2PResNetBlock(ch * 2 ** n_downsampling))
        for i in range(n_downsampling):
            m = 2 ** (n_downsampling - i)
            self.main.append(nn.Sequential(nn.ConvTranspose2d(ch * m, ch * m // 2, kernel_size=3, stride=2, padding=1, output_padding=1),
                                           nn.BatchNorm2d(ch * m // 2), nn.ReLU()))
        self.main.append(nn.Sequential(nn.ReflectionPad2d(3), nn.Conv2d(ch, output_dim, kernel_size=7, padding=0), nn.Tanh()))

    def forward(self, x, dropout_masks=None):
        out, bi = x, 0
        for layer in self.main:
            if isinstance(layer, P2PResNetBlock):
                out = layer(out, None if dropout_masks is None else dropout_masks[bi])
                bi += 1
            else:
                out = layer(out)
        return out


def _set_requires_grad(model, flag):
    for p in model.parameters():
        p.requires_grad = flag


def pix2pix_discriminator_phase(net_G, net_D, real, label, dropout_masks=None):
    bce = torch.nn.functional.binary_cross_entropy
    _set_requires_grad(net_G, False); _set_requires_grad(net_D, True)
    d_x = net_D(real, label)
    d_x = bce(d_x, torch.ones_like(d_x))
    fake = net_G(label, dropout_masks)
    d_g_z1 = net_D(fake.detach(), label)
    d_g_z1 = bce(d_g_z1, torch.zeros_like(d_g_z1))
    loss = (d_x + d_g_z1) * 0.5
    loss.backward()
    return loss.detach()


def pix2pix_generator_phase(net_G, net_D, real, label, lmbd=100.0, dropout_masks=None):
    bce = torch.nn.functional.binary_cross_entropy
    _set_requires_grad(net_G, True); _set_requires_grad(net_D, False)
    fake = net_G(label, dropout_masks)
    d_g_z2 = net_D(fake, label)
    g_error = bce(d_g_z2, torch.ones_like(d_g_z2)) + lmbd * torch.nn.functional.l1_loss(fake, real)
    g_error.backward()
    _set_requires_grad(net_G, False); _set_requires_grad(net_D, True)
    return g_error.detach(), fake.detach()
