"""One TRAINING step of each model family at the reference's per-GPU size, on synthetic data: both phases with their
optimizer steps (and the generator's moving average where the reference keeps one), as a callable that can be timed
eagerly, recorded into a HIP graph (gan_amd.GraphedStep, one rank) or run data-parallel (one process per GPU, the
gradient buckets of gan_amd.ddp.GradReducer all-reduced while the rest of the backward pass still runs).  The speed
options of the harnesses are on: bf16 activation storage, fused normalisation.  Used by bench.py and the tools.

  dcgan       BasicGANs/DCGAN.py:165-181      64x64, hidden 64, nz 100, batch 128; Adam 2e-4, betas (0.5, 0.999)
  pix2pix     Pix2Pix/train.py:94-145         U-Net + PatchGAN (ch 64), 256x256, batch 8 (the per-GPU share of 64 on 8 GPUs)
              and batch 64; Adam 2e-4, betas (0.5, 0.999) for both networks (Pix2Pix/cfg.py)
  stylegan    StyleGAN/train.py:211-248       progressive step 6 (256x256), batch 32, style mixing, gradient penalty; Adam
              2e-3 (cfg.lr[256]), betas (0, 0.9), the mapping network at 0.01 x that rate (train.py:134-143), moving average
              of the generator with weight 0.999 after its step (train.py:71-76, 244)
  biggan      BigGAN/train.py:187-226         128x128, ch 64, batch 256; Adam 2e-4 / 5e-5, betas (0, 0.999), orthogonal
              regularisation, moving average 0.9999.  The reference updates the generator on every second iteration
              (cfg.disc_repeats = 2); a step here is one discriminator update AND one generator update.

Every step keeps the loss terms of its phases in `step.losses` (a device tensor; bench.py checks that they are finite).
"""
import os

import torch

from .. import ops, optim
from . import basic_gans, biggan, pix2pix, stylegan


def _reseed(data_seed):
    """The networks are initialised from seed 5000 on every rank; the synthetic batch of a rank comes from its own seed."""
    if data_seed is not None:
        torch.manual_seed(data_seed)


class TrainStep:
    """step() = discriminator phase, its Adam step, generator phase, its Adam step.  With `reducer_factory`
    (gan_amd.ddp.GradReducer) every network's gradients live in flat buckets that are all-reduced over the ranks: a bucket's
    collective starts from the autograd hook of its last parameter during the phase's last backward pass."""

    def __init__(self, nets, opts, phases, batch, what, n_losses, device, reducer_factory=None, after=None):
        self.net_D, self.net_G = nets
        self.opt_D, self.opt_G = opts
        self.phase_D, self.phase_G = phases
        self.batch, self.what = batch, what
        self.losses = torch.zeros(n_losses, device=device)
        self.red_D = reducer_factory(self.net_D) if reducer_factory is not None else None
        self.red_G = reducer_factory(self.net_G) if reducer_factory is not None else None
        self.after = after or (lambda which: None)     # what follows a phase's gradients (BigGAN: orthogonal regularisation)
        # Both optimizers refresh the packed weight images of their network right after their step (one launch per network):
        # a recording of the step may then keep the images of the warm-up, which every replay finds current.
        self.weight_images_refreshed_in_step = all(getattr(o, "_nets", None) for o in opts)

    def _run(self, red, opt, phase, which):
        if red is not None:
            red.zero_grad()
            out = phase(red.arm)
            red.finish()
        else:
            opt.zero_grad(set_to_none=True)      # the first backward pass of a phase then writes its gradients instead of adding
            out = phase(None)
        self.after(which)
        opt.step()
        return out

    def __call__(self):
        d = self._run(self.red_D, self.opt_D, self.phase_D, "D")
        g = self._run(self.red_G, self.opt_G, self.phase_G, "G")
        torch.stack([t.detach().float().reshape(()) for t in (*d, *g)], out=self.losses)
        ops.clear_cast_cache()

    def parameters(self):
        return list(self.net_G.parameters()) + list(self.net_D.parameters())

    def buffers(self):
        return list(self.net_G.buffers()) + list(self.net_D.buffers())


def dcgan(device, batch=128, reducer_factory=None, data_seed=None):
    torch.manual_seed(5000)
    G = basic_gans.Generator(100, 64, 3, 64, activation_dtype=torch.bfloat16).to(device)
    D = basic_gans.Discriminator(3, 64, 64, is_wgan=False, activation_dtype=torch.bfloat16).to(device)
    G.apply(basic_gans.weights_init)
    D.apply(basic_gans.weights_init)
    oG = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999)).repack_after_step(G)
    oD = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)).repack_after_step(D)
    _reseed(data_seed)
    real = (torch.rand(batch, 3, 64, 64, device=device) * 2 - 1).contiguous(memory_format=torch.channels_last)
    z = torch.randn(batch, 100, 1, 1, device=device)
    keep = {}

    def phase_D(arm):
        err, d_fake, _, keep["fake"] = basic_gans.dcgan_discriminator_phase(G, D, real, z, on_last_backward=arm)
        return err, d_fake

    def phase_G(arm):
        # (freeze_critic=True would skip the critic gradients this phase throws away; kept as the reference computes them)
        return basic_gans.dcgan_generator_phase(D, keep.pop("fake"), on_last_backward=arm)[:1]

    return TrainStep((D, G), (oD, oG), (phase_D, phase_G), batch,
                     f"DCGAN 64x64 (hidden 64, nz 100), batch {batch}: discriminator + generator update with Adam", 3, device, reducer_factory)


def pix2pix_unet(device, batch=8, ch=64, size=256, reducer_factory=None, data_seed=None):
    torch.manual_seed(5000)
    pix2pix.set_activation_dtype(torch.bfloat16)
    pix2pix.set_fused_norm(True)
    G, D = pix2pix.Generator(ch=ch).to(device), pix2pix.Discriminator(ch=ch).to(device)
    oG = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999)).repack_after_step(G)
    oD = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999)).repack_after_step(D)
    _reseed(data_seed)
    # (channels-last, as gan_amd.data.ImageBatcher hands its batches over: the pair the critic reads is then built in the
    #  convolutions' own layout)
    real = (torch.rand(batch, 3, size, size, device=device) * 2 - 1).contiguous(memory_format=torch.channels_last)
    label = torch.rand(batch, 3, size, size, device=device).contiguous(memory_format=torch.channels_last)

    def phase_D(arm):
        return (pix2pix.discriminator_phase(G, D, real, label, on_last_backward=arm),)

    def phase_G(arm):
        return (pix2pix.generator_phase(G, D, real, label, on_last_backward=arm)[0],)

    return TrainStep((D, G), (oD, oG), (phase_D, phase_G), batch,
                     f"Pix2Pix U-Net + PatchGAN (ch {ch}), {size}x{size}, batch {batch}: discriminator + generator update with Adam",
                     2, device, reducer_factory)


def pix2pix_unet_b64(device, reducer_factory=None, data_seed=None):
    return pix2pix_unet(device, batch=64, reducer_factory=reducer_factory, data_seed=data_seed)


def stylegan_v1(device, batch=32, pg=6, reducer_factory=None, data_seed=None):
    torch.manual_seed(5000)
    stylegan.set_activation_dtype(torch.bfloat16)
    G, D = stylegan.StyleBasedGenerator().to(device), stylegan.Discriminator().to(device)
    averaged_G = stylegan.StyleBasedGenerator().to(device)
    averaged_G.load_state_dict(G.state_dict())          # moving_average(averaged_G, net_G, weight=0), StyleGAN/train.py:132
    averaged_G.train(False)
    res = 4 * 2 ** pg
    lr = {4: 1e-3, 8: 1e-3, 16: 1e-3, 32: 1e-3, 64: 1e-3, 128: 1.5e-3, 256: 2e-3, 512: 3e-3, 1024: 3e-3}[res]      # StyleGAN/cfg.py:17
    oG = optim.Adam([{"params": list(G.synthesis_network.parameters())},
                     {"params": list(G.mapping_network.parameters()), "lr": lr * 0.01}], lr=lr, betas=(0.0, 0.9))
    oG.average_into(averaged_G, G, 0.999)
    oG.repack_after_step(G)
    oD = optim.Adam(D.parameters(), lr=lr, betas=(0.0, 0.9)).repack_after_step(D)
    _reseed(data_seed)
    real = (torch.rand(batch, 3, res, res, device=device) * 2 - 1).contiguous(memory_format=torch.channels_last)
    zs = torch.randn(4, batch, 512, device=device)
    eps = torch.rand(batch, 1, 1, 1, device=device)
    cross = min(3, pg)

    def phase_D(arm):
        return stylegan.critic_phase(G, D, real, [zs[0], zs[1]], eps, pg, crossover_point=cross, on_last_backward=arm)[:3]

    def phase_G(arm):
        return (stylegan.generator_phase(G, D, [zs[2], zs[3]], pg, crossover_point=cross, on_last_backward=arm)[0],)

    step = TrainStep((D, G), (oD, oG), (phase_D, phase_G), batch,
                     f"StyleGAN-v1 progressive step {pg} ({res}x{res}), batch {batch}: critic update (gradient penalty) + generator update "
                     "(style mixing) with Adam, moving average of the generator", 4, device, reducer_factory)
    step.averaged_G = averaged_G
    return step


def biggan_128(device, batch=256, ch=64, n_classes=10, reducer_factory=None, data_seed=None):
    torch.manual_seed(5000)
    biggan.set_activation_dtype(torch.bfloat16)
    biggan.set_fused_attention(True)
    # (GC_BIGGAN_INIT_ON_CPU: the QR factorisations of the orthogonal initialisation on the host, for a profiler whose counter
    #  passes do not survive the solver library's kernels; other values than the device's QR, the same distribution)
    init_device = "cpu" if os.environ.get("GC_BIGGAN_INIT_ON_CPU") else device
    G, D = biggan.Generator(n_classes=n_classes, ch=ch).to(init_device), biggan.Discriminator(n_classes=n_classes, ch=ch).to(init_device)
    biggan.orthogonal_init(G)
    biggan.orthogonal_init(D)
    G, D = G.to(device), D.to(device)
    averaged_G = biggan.Generator(n_classes=n_classes, ch=ch).to(device)
    averaged_G.load_state_dict(G.state_dict())
    averaged_G.train(False)
    oG = optim.Adam(G.parameters(), lr=2e-4, betas=(0.0, 0.999))          # BigGAN/cfg.py
    oG.average_into(averaged_G, G, 0.9999)                                # BigGAN/train.py:83
    oD = optim.Adam(D.parameters(), lr=5e-5, betas=(0.0, 0.999))
    _reseed(data_seed)
    real = (torch.rand(batch, 3, 128, 128, device=device) * 2 - 1).contiguous(memory_format=torch.channels_last)
    z1, z2 = torch.rand(batch, 120, device=device), torch.rand(batch, 120, device=device)      # torch.rand: BigGAN/train.py:182
    lab = torch.randint(0, n_classes, (3, batch), device=device)

    def phase_D(arm):
        return biggan.discriminator_phase(G, D, real, lab[0], z1, lab[1], orth_reg=False, on_last_backward=arm)

    def phase_G(arm):
        return (biggan.generator_phase(G, D, z2, lab[2], orth_reg=False, on_last_backward=arm),)

    def after(which):          # on the reduced gradients: the term depends on the weights only, every rank adds the same
        biggan.orthogonal_regularization(D if which == "D" else G)

    step = TrainStep((D, G), (oD, oG), (phase_D, phase_G), batch,
                     f"BigGAN {128}x{128} (ch {ch}, {n_classes} classes), batch {batch}: discriminator update + generator update (hinge loss, "
                     "orthogonal regularisation) with Adam, moving average of the generator", 3, device, reducer_factory, after)
    step.averaged_G = averaged_G
    # every packed image this step caches belongs to a spectrally normalised convolution and is refreshed in front of each
    # forward pass (ops.repack_divided): a recording may keep the images of the warm-up
    step.weight_images_refreshed_in_step = bool(biggan._opt["divided_pack"]) and os.environ.get("GC_BIGGAN_CLEAR_IMAGES") is None
    return step


FAMILIES = {"dcgan64_b128": dcgan, "pix2pix256_b8": pix2pix_unet, "pix2pix256_b64": pix2pix_unet_b64, "stylegan256_b32": stylegan_v1,
            "biggan128_b256": biggan_128}


def reset_options():
    pix2pix.set_activation_dtype(None)
    stylegan.set_activation_dtype(None)
    biggan.set_activation_dtype(None)
