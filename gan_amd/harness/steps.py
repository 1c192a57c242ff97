"""One training step of each further model family at the reference's per-GPU size, on synthetic data, as a callable that
can be timed eagerly or recorded into a HIP graph (gan_amd.GraphedStep).  The speed options of the harnesses are on: bf16
activation storage, fused normalisation.  Used by bench.py's `families` figures and tools/family_bench.py.

  dcgan      BasicGANs/DCGAN.py:165-181      64x64, hidden 64, nz 100, batch 128, both Adam steps
  pix2pix    Pix2Pix/train.py (U-Net + PatchGAN, ch 64)   256x256, batch 8, gradients of both phases
  stylegan   StyleGAN/train.py               progressive step 6 (256x256), batch 32, style mixing, gradient penalty
"""
import torch

from .. import ops, optim
from . import basic_gans, pix2pix, stylegan


def _zero(*nets):
    for n in nets:
        for p in n.parameters():
            p.grad = None


def dcgan(device, batch=128):
    torch.manual_seed(5000)
    G = basic_gans.Generator(100, 64, 3, 64, activation_dtype=torch.bfloat16).to(device)
    D = basic_gans.Discriminator(3, 64, 64, is_wgan=False, activation_dtype=torch.bfloat16).to(device)
    G.apply(basic_gans.weights_init)
    D.apply(basic_gans.weights_init)
    oG = optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    oD = optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    real = (torch.rand(batch, 3, 64, 64, device=device) * 2 - 1).contiguous(memory_format=torch.channels_last)
    z = torch.randn(batch, 100, 1, 1, device=device)

    def step():
        oD.zero_grad(set_to_none=True)       # (the first backward pass of a phase then writes its gradients instead of adding)
        *_, fake = basic_gans.dcgan_discriminator_phase(G, D, real, z)
        oD.step()
        oG.zero_grad(set_to_none=True)
        basic_gans.dcgan_generator_phase(D, fake)      # (freeze_critic=True would skip the critic gradients this phase throws away: 1.84 instead of 2.01 ms; kept as the reference computes them)
        oG.step()
        ops.clear_cast_cache()

    return step, batch, f"DCGAN 64x64 (hidden 64, nz 100), batch {batch}: discriminator + generator update with Adam"


def pix2pix_unet(device, batch=8):
    torch.manual_seed(5000)
    pix2pix.set_activation_dtype(torch.bfloat16)
    pix2pix.set_fused_norm(True)
    G, D = pix2pix.Generator(ch=64).to(device), pix2pix.Discriminator(ch=64).to(device)
    real = torch.rand(batch, 3, 256, 256, device=device) * 2 - 1
    label = torch.rand(batch, 3, 256, 256, device=device)

    def step():
        _zero(G, D)
        pix2pix.discriminator_phase(G, D, real, label)
        pix2pix.generator_phase(G, D, real, label)
        ops.clear_cast_cache()

    return step, batch, f"Pix2Pix U-Net + PatchGAN (ch 64), 256x256, batch {batch}: gradients of both phases"


def stylegan_v1(device, batch=32, pg=6):
    torch.manual_seed(5000)
    stylegan.set_activation_dtype(torch.bfloat16)
    G, D = stylegan.StyleBasedGenerator().to(device), stylegan.Discriminator().to(device)
    res = 4 * 2 ** pg
    real = torch.rand(batch, 3, res, res, device=device) * 2 - 1
    zs = torch.randn(4, batch, 512, device=device)
    eps = torch.rand(batch, 1, 1, 1, device=device)

    def step():
        _zero(G, D)
        stylegan.critic_phase(G, D, real, [zs[0], zs[1]], eps, pg, crossover_point=3)
        stylegan.generator_phase(G, D, [zs[2], zs[3]], pg, crossover_point=3)
        ops.clear_cast_cache()

    return step, batch, f"StyleGAN-v1 progressive step {pg} ({res}x{res}), batch {batch}: critic (gradient penalty) + generator phase, style mixing"


FAMILIES = {"dcgan64_b128": dcgan, "pix2pix256_b8": pix2pix_unet, "stylegan256_b32": stylegan_v1}


def reset_options():
    pix2pix.set_activation_dtype(None)
    stylegan.set_activation_dtype(None)
