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
