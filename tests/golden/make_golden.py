"""Generate the golden vectors under tests/golden/ from the reference checkout.

Runs ONLY in the build container (needs /root/reference, CPU torch).  Nothing on the GPU box reads the
reference: the .npz / .json files this script writes are committed and are the only thing that travels.

What it captures, always with the reference's seed (5000) and the reference's own model / loss files
imported by path (BasicGANs/models.py, BasicGANs/loss_functions.py):

  ops_*.npz            known-answer vectors per convolution geometry class, produced with the call forms
                       the reference uses (torch.nn.Conv2d / ConvTranspose2d / F.conv2d / F.conv_transpose2d)
  dcgan_trace.npz      one DCGAN G+D step (get_minimax_loss, step order of BasicGANs/DCGAN.py:165-181) on a
                       reduced model (hidden_dim 8, batch 2): every conv layer's input, weight, output,
                       grad-output, dgrad and the parameter gradients
  wgangp_trace.npz     one WGAN-GP critic iteration + one generator iteration (get_wasserstein_loss,
                       BasicGANs/train.py:95-119) incl. the gradient-penalty double backward, reduced model
  dcgan_losses.json    loss scalars of 3 full-size DCGAN-64 steps (hidden 64, batch 128) with Adam
  wgangp_losses.json   loss scalars of 1 full cycle (5 critic + 1 generator) WGAN-GP-64 at batch 64 with Adam
  pix2pixhd_trace.npz  forward + backward of the Pix2PixHD global generator (Pix2Pix/models.py:150-218), reduced size
  pix2pix_trace.npz    one Pix2Pix step (Pix2Pix/models.py, step order of Pix2Pix/train.py:94-145) at 256x256, ch 8,
                       batch 2: losses, discriminator gradients, sampled generator gradients, output crops, masks
  stylegan_trace.npz   StyleGAN-v1 (StyleGAN/model.py) at 128x128 (pg step 5), batch 2: critic phase with style mixing and
                       gradient penalty, generator phase, a fade-in forward pass; all internal random draws recorded
  biggan_trace.npz     BigGAN (BigGAN/model.py) at 128x128 with ch 8, 4 classes, batch 4: hinge discriminator and generator
                       phases with orthogonal init / regularisation

Usage:  python tests/golden/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import hashlib
import importlib.util
import json
import os
import random
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.dont_write_bytecode = True  # the reference tree is read-only
SEED = 5000


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def seed_all(seed=SEED):
    random.seed(seed)
    torch.manual_seed(seed)
    np.random.seed(seed)


def weights_init(m):
    # BasicGANs/utils.py:31-36 (that file imports cv2, which is not installed, so it is restated here)
    if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
    if isinstance(m, torch.nn.BatchNorm2d):
        torch.nn.init.normal_(m.weight, 0.0, 0.02)
        torch.nn.init.constant_(m.bias, 0)


def checksum(t):
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]


# ----------------------------------------------------------------------------------------------- ops
OP_CASES = [
    # name, kind, N, Cin, H, W, Cout, k, stride, pad, bias, output_padding
    ("conv_k4s2p1_c3", "conv", 2, 3, 16, 16, 16, 4, 2, 1, False, 0),       # BasicGANs D first layer
    ("conv_k4s2p1_c32", "conv", 2, 32, 16, 16, 64, 4, 2, 1, True, 0),      # BasicGANs / Pix2Pix / PatchGAN
    ("conv_k4s1p0_c64", "conv", 3, 64, 4, 4, 1, 4, 1, 0, False, 0),        # BasicGANs D last layer (4x4 -> 1x1)
    ("conv_k4s1p1_c16", "conv", 2, 16, 9, 9, 24, 4, 1, 1, True, 0),        # PatchGAN conv_4/5 (odd output size)
    ("conv_k3s1p1_c24", "conv", 2, 24, 10, 12, 40, 3, 1, 1, True, 0),      # StyleGAN / BigGAN 3x3
    ("conv_k1s1p0_c40", "conv", 2, 40, 7, 5, 3, 1, 1, 0, True, 0),         # toRGB / fromRGB / BigGAN skip 1x1
    ("conv_k3s1p1_c513", "conv", 1, 513, 4, 4, 8, 3, 1, 1, True, 0),       # StyleGAN D minibatch-std layer
    ("convt_k4s1p0_c100", "convt", 3, 100, 1, 1, 32, 4, 1, 0, False, 0),   # BasicGANs G first layer (1x1 -> 4x4)
    ("convt_k4s2p1_c32", "convt", 2, 32, 8, 8, 16, 4, 2, 1, True, 0),      # BasicGANs G / Pix2Pix decoder / FusedUpsample
    ("convt_k4s2p1_c16_o3", "convt", 2, 16, 8, 8, 3, 4, 2, 1, False, 0),   # BasicGANs G last layer (Cout 3)
    ("convt_k3s2p1_op1", "convt", 2, 8, 5, 5, 8, 3, 2, 1, True, 1),        # Pix2PixHD upsampling (output_padding)
]


def make_ops(out):
    seed_all()
    blob = {}
    for (name, kind, N, Cin, H, W, Cout, k, s, p, bias, op) in OP_CASES:
        if kind == "conv":
            m = torch.nn.Conv2d(Cin, Cout, k, s, p, bias=bias)
        else:
            m = torch.nn.ConvTranspose2d(Cin, Cout, k, s, p, output_padding=op, bias=bias)
        torch.nn.init.normal_(m.weight, 0.0, 0.5)
        x = torch.randn(N, Cin, H, W, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        grads = torch.autograd.grad(y, [x, m.weight] + ([m.bias] if bias else []), gy)
        blob[name + ".x"] = x.detach().numpy()
        blob[name + ".w"] = m.weight.detach().numpy()
        if bias:
            blob[name + ".b"] = m.bias.detach().numpy()
            blob[name + ".gb"] = grads[2].numpy()
        blob[name + ".y"] = y.detach().numpy()
        blob[name + ".gy"] = gy.numpy()
        blob[name + ".gx"] = grads[0].numpy()
        blob[name + ".gw"] = grads[1].numpy()
        blob[name + ".meta"] = np.array([N, Cin, H, W, Cout, k, s, p, int(bias), op, int(kind == "convt")], np.int64)
    # depthwise blur as the reference calls it (StyleGAN/model.py:95: F.conv2d(x, kernel, padding=1, groups=C))
    C = 6
    kern = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.0
    kern = kern.view(1, 1, 3, 3).repeat(C, 1, 1, 1)
    x = torch.randn(2, C, 9, 7)
    blob["blur.x"] = x.numpy()
    blob["blur.y"] = F.conv2d(x, kern, padding=1, groups=C).numpy()
    np.savez_compressed(os.path.join(out, "ops.npz"), **blob)
    print("ops.npz:", len(OP_CASES), "cases")


# ------------------------------------------------------------------------------------- full-width layers
# Single layers at the reference's full channel widths (the reduced traces above run at ch 8): inputs are regenerated from
# a torch CPU generator seed by the test, the file holds the results of the reference's call forms.  Large results are
# stored as a strided sample plus eight random +-1 projections of the whole tensor.
WIDE_CASES = [
    # name, kind, N, Cin, H, W, Cout, k, stride, pad
    ("p2p_enc_512_4to2", "conv", 4, 512, 4, 4, 512, 4, 2, 1),       # Pix2Pix/models.py:225-232 (U-Net bottom, ch 64)
    ("p2p_enc_512_2to1", "conv", 4, 512, 2, 2, 512, 4, 2, 1),
    ("p2p_dec_512_1to2", "convt", 4, 512, 1, 1, 512, 4, 2, 1),
    ("p2p_dec_1024_2to4", "convt", 4, 1024, 2, 2, 512, 4, 2, 1),    # (skip connection doubles the input channels)
    ("patchgan_256_512_s1", "conv", 2, 256, 31, 31, 512, 4, 1, 1),  # Pix2Pix/models.py:123-124 (PatchGAN conv_4 at 256x256 input)
    ("stylegan_pg6_3x3", "conv", 1, 64, 256, 256, 32, 3, 1, 1),     # StyleGAN/model.py:72 (synthesis block 6, 256x256)
    ("biggan_1024_8x8", "conv", 2, 1024, 8, 8, 1024, 3, 1, 1),      # BigGAN/model.py:121-124 (generator block 1 at ch 64)
]
WIDE_SAMPLE = 32768


def wide_inputs(case, seed=SEED):
    """(x, w, gy) of a WIDE_CASES entry, from a CPU generator (the same numbers wherever this torch version runs)."""
    name, kind, N, Cin, H, W, Cout, k, s, p = case
    g = torch.Generator().manual_seed(seed + sum(name.encode()))
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn((Cout, Cin, k, k) if kind == "conv" else (Cin, Cout, k, k), generator=g) * (2.0 / (Cin * k * k)) ** 0.5
    Ho = (H + 2 * p - k) // s + 1 if kind == "conv" else (H - 1) * s - 2 * p + k
    Wo = (W + 2 * p - k) // s + 1 if kind == "conv" else (W - 1) * s - 2 * p + k
    gy = torch.randn(N, Cout, Ho, Wo, generator=g)
    return x, w, gy


def wide_digest(t, seed):
    """(strided sample of at most WIDE_SAMPLE elements, eight +-1 projections) of a tensor, in float64 accumulation."""
    flat = t.detach().reshape(-1)
    step = max(1, flat.numel() // WIDE_SAMPLE) | 1        # odd: walks through every tap and channel residue
    g = torch.Generator().manual_seed(seed)
    proj = []
    for _ in range(8):
        signs = torch.randint(0, 2, (flat.numel(),), generator=g, dtype=torch.int8).double() * 2 - 1
        proj.append(float((flat.double() * signs).sum()))
    return flat[::step].clone().numpy(), np.array(proj), float(flat.double().norm())


def make_wide_ops(out):
    blob = {}
    for case in WIDE_CASES:
        name, kind, N, Cin, H, W, Cout, k, s, p = case
        x, w, gy = wide_inputs(case)
        x.requires_grad_(True); w.requires_grad_(True)
        y = F.conv2d(x, w, None, s, p) if kind == "conv" else F.conv_transpose2d(x, w, None, s, p)
        gx, gw = torch.autograd.grad(y, [x, w], gy)
        for tag, t in (("y", y), ("gx", gx), ("gw", gw)):
            sample, proj, norm = wide_digest(t, 17)
            blob[f"{name}.{tag}.sample"], blob[f"{name}.{tag}.proj"], blob[f"{name}.{tag}.norm"] = sample, proj, np.array(norm)
        blob[name + ".meta"] = np.array([N, Cin, H, W, Cout, k, s, p, int(kind == "convt")], np.int64)
    np.savez_compressed(os.path.join(out, "wide_ops.npz"), **blob)
    print("wide_ops.npz:", len(WIDE_CASES), "cases")


# --------------------------------------------------------------------------------------------- traces
class ConvTap:
    """Records (x, y, gy, gx) of every Conv2d / ConvTranspose2d call, in call order."""

    def __init__(self, net, tag):
        self.records = []
        self.tag = tag
        for name, mod in net.named_modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                mod.register_forward_hook(self._hook(name))

    def _hook(self, name):
        def fn(mod, inp, out):
            rec = {"name": name, "x": inp[0].detach().clone(), "y": out.detach().clone()}
            self.records.append(rec)
            if out.requires_grad:
                out.register_hook(lambda g, rec=rec: rec.__setitem__("gy", g.detach().clone()))
            if inp[0].requires_grad:
                inp[0].register_hook(lambda g, rec=rec: rec.__setitem__("gx", g.detach().clone()))
        return fn

    def dump(self, blob, prefix):
        for i, rec in enumerate(self.records):
            for key in ("x", "y", "gy", "gx"):
                if key in rec:
                    blob[f"{prefix}.{self.tag}.call{i:02d}.{rec['name']}.{key}"] = rec[key].numpy()


def make_dcgan_trace(ref, out):
    models = load_by_path("ref_basic_models", os.path.join(ref, "BasicGANs/models.py"))
    losses = load_by_path("ref_basic_losses", os.path.join(ref, "BasicGANs/loss_functions.py"))
    seed_all()
    B, hd, nz, res = 2, 8, 100, 64
    net_G = models.Generator(nz, hd, 3, resolution=res); net_G.apply(weights_init)
    net_D = models.Discriminator(3, hd, resolution=res); net_D.apply(weights_init)
    blob = {}
    for k, v in net_G.state_dict().items():
        blob["init.G." + k] = v.clone().numpy()
    for k, v in net_D.state_dict().items():
        blob["init.D." + k] = v.clone().numpy()
    real = torch.rand(B, 3, res, res) * 2 - 1
    z = torch.randn(B, nz, 1, 1)
    blob["real"] = real.numpy(); blob["z"] = z.numpy()
    tapG, tapD = ConvTap(net_G, "G"), ConvTap(net_D, "D")
    criterion = torch.nn.BCELoss()
    # step order: BasicGANs/DCGAN.py:165-181
    net_D.zero_grad()
    label = torch.full((B,), 1.0)
    D_error, D_G_z1, D_x, fake = losses.get_minimax_loss(net_G, net_D, criterion, label, real, z, 0.0)
    for k, p in net_D.named_parameters():
        blob["dphase.gradD." + k] = p.grad.clone().numpy()
    blob["dphase.scalars"] = np.array([D_error.item(), D_G_z1, D_x], np.float64)
    net_G.zero_grad()
    label.fill_(1.0)
    G_error, D_G_z2 = losses.get_minimax_loss(net_G, net_D, criterion, label, fake, discriminator=False)
    for k, p in net_G.named_parameters():
        blob["gphase.gradG." + k] = p.grad.clone().numpy()
    blob["gphase.scalars"] = np.array([G_error.item(), D_G_z2], np.float64)
    blob["fake"] = fake.detach().numpy()
    tapG.dump(blob, "trace"); tapD.dump(blob, "trace")
    for k, v in net_G.state_dict().items():   # BatchNorm running statistics after the step
        if "running" in k:
            blob["after.G." + k] = v.clone().numpy()
    for k, v in net_D.state_dict().items():
        if "running" in k:
            blob["after.D." + k] = v.clone().numpy()
    np.savez_compressed(os.path.join(out, "dcgan_trace.npz"), **blob)
    print("dcgan_trace.npz:", len(blob), "arrays; scalars", blob["dphase.scalars"], blob["gphase.scalars"])


def make_wgangp_trace(ref, out):
    models = load_by_path("ref_basic_models", os.path.join(ref, "BasicGANs/models.py"))
    losses = load_by_path("ref_basic_losses", os.path.join(ref, "BasicGANs/loss_functions.py"))
    seed_all()
    B, hd, nz, res = 4, 8, 100, 64
    net_G = models.Generator(nz, hd, 3, resolution=res); net_G.apply(weights_init)
    net_D = models.Discriminator(3, hd, resolution=res, is_wgan=True); net_D.apply(weights_init)
    blob = {}
    for k, v in net_G.state_dict().items():
        blob["init.G." + k] = v.clone().numpy()
    for k, v in net_D.state_dict().items():
        blob["init.D." + k] = v.clone().numpy()
    real = torch.rand(B, 3, res, res) * 2 - 1
    z = torch.randn(B, nz, 1, 1)
    z1 = torch.randn(B, nz, 1, 1)
    blob["real"] = real.numpy(); blob["z"] = z.numpy(); blob["z1"] = z1.numpy()
    # capture the epsilon the loss function draws internally (loss_functions.py:18)
    drawn = []
    orig_rand = torch.rand

    def rand_spy(*a, **kw):
        t = orig_rand(*a, **kw)
        drawn.append(t.detach().clone())
        return t
    torch.rand = rand_spy
    try:
        for p in net_D.parameters():
            p.grad = None
        D_x, D_G_z1, gp = losses.get_wasserstein_loss(net_G, net_D, real, z)
    finally:
        torch.rand = orig_rand
    blob["epsilon"] = drawn[0].numpy()
    blob["critic.scalars"] = np.array([D_x.item(), D_G_z1, gp.item()], np.float64)
    for k, p in net_D.named_parameters():
        blob["critic.gradD." + k] = p.grad.clone().numpy()
    # the first-order gradient w.r.t. the mixed images, recomputed the way the loss does (loss_functions.py:19-25)
    eps = drawn[0].clone().requires_grad_(True)
    with torch.no_grad():
        fake = net_G(z)
    mixed = real * eps + fake * (1 - eps)
    g = torch.autograd.grad(outputs=net_D(mixed).sum(), inputs=mixed, create_graph=False)[0]
    blob["critic.mixed_grad"] = g.numpy()
    blob["fake"] = fake.numpy()
    for p in net_G.parameters():
        p.grad = None
    for p in net_D.parameters():
        p.grad = None
    G_error, D_G_z2 = losses.get_wasserstein_loss(net_G, net_D, noise_vector=z1, discriminator=False)
    blob["gen.scalars"] = np.array([G_error.item(), D_G_z2], np.float64)
    for k, p in net_G.named_parameters():
        blob["gen.gradG." + k] = p.grad.clone().numpy()
    np.savez_compressed(os.path.join(out, "wgangp_trace.npz"), **blob)
    print("wgangp_trace.npz:", len(blob), "arrays; scalars", blob["critic.scalars"], blob["gen.scalars"])


def make_pix2pix_trace(ref, out, B=2, ch=8, name="pix2pix_trace"):
    """One Pix2Pix step (Pix2Pix/train.py:94-145) on the reference's models at 256x256 with ch=8, batch 2
    (`pix2pix_trace`) and at the reference's full width, ch=64, batch 1 (`pix2pix_wide_trace`: every gradient as a strided
    sample, its norm and eight +-1 projections of the whole tensor; dropout masks as seeds).  The
    nets keep PyTorch's default initialisation (the script applies none), regenerated from the seed by the tests
    and pinned here by checksums; dropout masks are captured."""
    models = load_by_path("ref_p2p_models", os.path.join(ref, "Pix2Pix/models.py"))
    seed_all()
    wide = ch > 8
    net_G = models.Generator(ch=ch)
    net_D = models.Discriminator(ch=ch)
    blob = {"cfg": np.array([B, ch], np.int64)}
    rec = {"param_count": {"G": sum(p.numel() for p in net_G.parameters()), "D": sum(p.numel() for p in net_D.parameters())},
           "init_checksum": {"G": checksum(torch.cat([p.flatten() for p in net_G.parameters()])),
                             "D": checksum(torch.cat([p.flatten() for p in net_D.parameters()]))}}
    gen = torch.Generator().manual_seed(SEED + 2)
    real = torch.rand(B, 3, 256, 256, generator=gen) * 2 - 1
    label = torch.rand(B, 3, 256, 256, generator=gen)
    rec["input_checksum"] = {"real": checksum(real), "label": checksum(label)}
    masks = []
    orig_dropout = F.dropout

    def dropout_spy(x, p=0.5, training=True, inplace=False):
        m = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype) / (1 - p)
        masks.append(m)
        return x * m
    F.dropout = dropout_spy
    torch.nn.functional.dropout = dropout_spy
    try:
        bce = torch.nn.BCELoss(); l1 = torch.nn.L1Loss()
        req = lambda m, f: [p.requires_grad_(f) for p in m.parameters()]
        req(net_G, False); req(net_D, True)
        D_x, _ = net_D(real, label)
        ones = torch.ones_like(D_x)
        D_x = bce(D_x, ones)
        fake = net_G(label)
        D_G_z1, _ = net_D(fake.detach(), label)
        D_G_z1 = bce(D_G_z1, torch.zeros_like(D_G_z1))
        d_loss = (D_x + D_G_z1) * 0.5
        d_loss.backward()
        blob["d.loss"] = np.array([d_loss.item(), D_x.item(), D_G_z1.item()])
        for k, p in net_D.named_parameters():
            blob["d.gradD." + k] = _sample(p.grad.clone().numpy()) if wide else p.grad.clone().numpy()
            if wide:
                blob["d.gradD_norm." + k] = np.array([np.linalg.norm(p.grad.numpy().astype(np.float64))])
                blob["d.gradD_proj." + k] = _projections(p.grad)
        blob["d.fake_crop"] = fake.detach()[:, :, 96:160, 96:160].numpy()
        n_d_masks = len(masks)
        req(net_G, True); req(net_D, False)
        fake1 = net_G(label)
        D_G_z2, _ = net_D(fake1, label)
        g_err = bce(D_G_z2, ones) + 100.0 * l1(fake1, real)
        g_err.backward()
        blob["g.loss"] = np.array([g_err.item()])
        blob["g.fake_crop"] = fake1.detach()[:, :, 96:160, 96:160].numpy()
        for k, p in net_G.named_parameters():      # small tensors in full, big ones as a strided sample
            g = p.grad.clone().numpy()
            blob["g.gradG." + k] = g if g.size <= 20000 else g.ravel()[::max(1, g.size // 4096)][:4096].copy()
            blob["g.gradG_norm." + k] = np.array([np.linalg.norm(g.astype(np.float64))])
            if wide:
                blob["g.gradG_proj." + k] = _projections(p.grad)
    finally:
        F.dropout = orig_dropout
        torch.nn.functional.dropout = orig_dropout
    for i, m in enumerate(masks):          # 0 / 2 masks: one bit per element
        blob[f"mask{i}"] = np.packbits(m.numpy().astype(bool).ravel()) if wide else m.numpy()
        blob[f"mask{i}.shape"] = np.array(m.shape, np.int64)
    blob["n_d_masks"] = np.array([n_d_masks])
    np.savez_compressed(os.path.join(out, name + ".npz"), **blob)
    json.dump(rec, open(os.path.join(out, name + ".json"), "w"), indent=1)
    print(name + ".npz:", len(blob), "arrays; losses", blob["d.loss"], blob["g.loss"], rec["param_count"])


def make_pix2pixhd_trace(ref, out):
    """Forward + backward of the Pix2PixHD global generator (Pix2Pix/models.py:150-218: reflection-padded 7x7 and 3x3
    convolutions, stride-2 3x3 convolutions, residual blocks with dropout, ConvTranspose2d k3 s2 p1 output_padding 1) at
    64x64 with ch = 8, two residual blocks, batch 2, under an L1 loss against a random target.  Default initialisation
    regenerated from the seed by the tests and pinned by a checksum; dropout masks captured."""
    models = load_by_path("ref_p2p_models", os.path.join(ref, "Pix2Pix/models.py"))
    seed_all()
    B, ch, blocks, res = 2, 8, 2, 64
    net = models.GeneratorHD(ch=ch, n_blocks=blocks)
    blob = {"cfg": np.array([B, ch, blocks, res], np.int64)}
    rec = {"param_count": sum(p.numel() for p in net.parameters()),
           "init_checksum": checksum(torch.cat([p.flatten() for p in net.parameters()])),
           "state_dict_keys": list(net.state_dict().keys())}
    gen = torch.Generator().manual_seed(SEED + 5)
    label = (torch.rand(B, 3, res, res, generator=gen) * 2 - 1).requires_grad_(True)
    target = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
    masks = []
    orig_dropout = F.dropout

    def dropout_spy(x, p=0.5, training=True, inplace=False):
        m = (torch.rand(x.shape, generator=gen) >= p).to(x.dtype) / (1 - p)
        masks.append(m)
        return x * m
    F.dropout = dropout_spy
    torch.nn.functional.dropout = dropout_spy
    try:
        fake = net(label)
        loss = torch.nn.L1Loss()(fake, target)
        loss.backward()
    finally:
        F.dropout = orig_dropout
        torch.nn.functional.dropout = orig_dropout
    blob["label"] = label.detach().numpy(); blob["target"] = target.numpy()
    blob["fake"] = fake.detach().numpy(); blob["loss"] = np.array([loss.item()])
    blob["grad_label"] = label.grad.numpy()
    for k, p in net.named_parameters():
        blob["grad." + k] = p.grad.clone().numpy()
    for k, b in net.named_buffers():
        if "running" in k:
            blob["buf." + k] = b.clone().numpy()
    for i, m in enumerate(masks):
        blob[f"mask{i}"] = m.numpy()
    np.savez_compressed(os.path.join(out, "pix2pixhd_trace.npz"), **blob)
    json.dump(rec, open(os.path.join(out, "pix2pixhd_trace.json"), "w"), indent=1)
    print("pix2pixhd_trace.npz:", len(blob), "arrays; loss", blob["loss"], rec["param_count"], "masks", len(masks))


def _sample(g, n=4096):
    g = np.asarray(g)
    return g if g.size <= 20000 else g.ravel()[::max(1, g.size // n)][:n].copy()


def _projections(t, seed=29, n=8):
    """n random +-1 projections of the whole tensor in float64: every element enters every number."""
    flat = t.detach().reshape(-1).double()
    g = torch.Generator().manual_seed(seed)
    return np.array([float((flat * (torch.randint(0, 2, (flat.numel(),), generator=g, dtype=torch.int8).double() * 2 - 1)).sum())
                     for _ in range(n)])


def make_stylegan_trace(ref, out, pg=5, name="stylegan_trace"):
    """StyleGAN-v1 at progressive-growing step 5 (128x128, the first resolution with the fused up/down-sampling
    blocks) or 6 (256x256, `stylegan256_trace`: the resolution the throughput figures are quoted at), batch 2: one critic phase with style mixing and gradient penalty, one generator phase, and a
    forward-only fade-in (alpha = 0.5) pass, on the reference's StyleGAN/model.py + BasicGANs/loss_functions.py.
    Weights are re-created from the seed by the tests (checksums here); every random draw made inside forward /
    the loss (noise images, crossover, epsilon) is recorded."""
    import random as _random
    models = load_by_path("ref_sg_models", os.path.join(ref, "StyleGAN/model.py"))
    losses = load_by_path("ref_basic_losses", os.path.join(ref, "BasicGANs/loss_functions.py"))
    seed_all()
    B, res = 2, 4 * 2 ** pg
    lo, hi = res // 2 - 32, res // 2 + 32          # the recorded image window
    net_G = models.StyleBasedGenerator(fused=True)
    net_D = models.Discriminator(fused=True, use_activations=True)
    rec = {"param_count": {"G": sum(p.numel() for p in net_G.parameters()), "D": sum(p.numel() for p in net_D.parameters())},
           "init_checksum": {"G": checksum(torch.cat([p.flatten() for p in net_G.parameters()])),
                             "D": checksum(torch.cat([p.flatten() for p in net_D.parameters()]))},
           "B": B, "pg_step": pg}
    gen = torch.Generator().manual_seed(SEED + 3)
    real = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
    zs = torch.randn(4, B, 512, generator=gen)
    rec["input_checksum"] = {"real": checksum(real), "zs": checksum(zs)}
    blob = {}
    drawn = {"randn": [], "rand": [], "randint": []}
    o_randn, o_rand, o_randint = torch.randn, torch.rand, _random.randint

    def randn_spy(*a, **kw):
        t = o_randn(*a, **kw); drawn["randn"].append(t.detach().clone()); return t

    def rand_spy(*a, **kw):
        t = o_rand(*a, **kw); drawn["rand"].append(t.detach().clone()); return t

    def randint_spy(a, b):
        v = o_randint(a, b); drawn["randint"].append(v); return v
    torch.randn, torch.rand = randn_spy, rand_spy
    models.random.randint = randint_spy
    req = lambda m, f: [p.requires_grad_(f) for p in m.parameters()]
    try:
        req(net_G, False); req(net_D, True)
        D_x, D_G_z1, gp = losses.get_wasserstein_loss(net_G, net_D, real, [zs[0], zs[1]], progressive_growing_phase=pg, alpha=1)
        blob["critic.scalars"] = np.array([D_x.item(), D_G_z1, gp.item()])
        for i, t in enumerate(drawn["randn"]):
            blob[f"critic.noise{i}"] = t.numpy()
        blob["critic.eps"] = drawn["rand"][0].numpy()
        blob["critic.crossover"] = np.array(drawn["randint"])
        for k, p in net_D.named_parameters():
            if p.grad is not None:
                blob["critic.gradD." + k] = _sample(p.grad.numpy())
                blob["critic.gradD_norm." + k] = np.array([np.linalg.norm(p.grad.numpy().astype(np.float64))])
        for k in drawn: drawn[k].clear()
        req(net_G, True); req(net_D, False)
        G_error, D_G_z2 = losses.get_wasserstein_loss(net_G, net_D, noise_vector=[zs[2], zs[3]], discriminator=False,
                                                      progressive_growing_phase=pg, alpha=1)
        blob["gen.scalars"] = np.array([G_error.item(), D_G_z2])
        for i, t in enumerate(drawn["randn"]):
            blob[f"gen.noise{i}"] = t.numpy()
        blob["gen.crossover"] = np.array(drawn["randint"])
        for k, p in net_G.named_parameters():
            if p.grad is not None:
                blob["gen.gradG." + k] = _sample(p.grad.numpy())
                blob["gen.gradG_norm." + k] = np.array([np.linalg.norm(p.grad.numpy().astype(np.float64))])
        for k in drawn: drawn[k].clear()
        with torch.no_grad():
            img = net_G(zs[0], progressive_growing_phase=pg, alpha=0.5)
            score = net_D(img, progressive_growing_phase=pg, alpha=0.5)
            img1 = net_G(zs[0], progressive_growing_phase=pg, alpha=1)
        for i, t in enumerate(drawn["randn"]):
            blob[f"fade.noise{i}"] = t.numpy()
        blob["fade.image_crop"] = img[:, :, lo:hi, lo:hi].numpy()
        blob["fade.score"] = score.numpy()
        blob["fade.image_alpha1_crop"] = img1[:, :, lo:hi, lo:hi].numpy()
    finally:
        torch.randn, torch.rand = o_randn, o_rand
        models.random.randint = o_randint
    np.savez_compressed(os.path.join(out, name + ".npz"), **blob)
    json.dump(rec, open(os.path.join(out, name + ".json"), "w"), indent=1)
    print(name + ".npz:", len(blob), "arrays;", blob["critic.scalars"], blob["gen.scalars"], rec["param_count"], blob["critic.crossover"], blob["gen.crossover"])


def make_biggan_trace(ref, out, B=4, ch=8, name="biggan_trace"):
    """BigGAN 128x128 with ch = 8, 4 classes, batch 4 (`biggan_trace`) and at the width the reference trains, ch = 64, batch 2
    (`biggan_wide_trace`, with eight +-1 projections of every whole gradient) on the reference's BigGAN/model.py: hinge discriminator phase
    and generator phase (BasicGANs/loss_functions.py:79-99) with orthogonal init and orthogonal regularisation
    restated from BigGAN/train.py:108-127 (that script imports cv2 / tensorboard and cannot be imported)."""
    models = load_by_path("ref_bg_models", os.path.join(ref, "BigGAN/model.py"))
    losses = load_by_path("ref_basic_losses", os.path.join(ref, "BasicGANs/loss_functions.py"))
    seed_all()
    ncls = 4
    wide = ch > 8
    net_G = models.Generator(n_classes=ncls, ch=ch)
    net_D = models.Discriminator(n_classes=ncls, ch=ch)

    def orthogonal_init(model):
        for module in model.modules():
            if isinstance(module, (torch.nn.Conv2d, torch.nn.Linear, torch.nn.Embedding)):
                torch.nn.init.orthogonal_(module.weight)

    def orth_reg(module, beta=1e-4):
        with torch.no_grad():
            for p in module.parameters():
                if len(p.shape) > 2:
                    w = p.view(p.shape[0], -1)
                    p.grad.data += beta * (2 * torch.mm(torch.mm(w, w.t()) * (1. - torch.eye(w.shape[0])), w)).view(p.shape)
    orthogonal_init(net_G); orthogonal_init(net_D)
    rec = {"B": B, "ch": ch, "n_classes": ncls,
           "param_count": {"G": sum(p.numel() for p in net_G.parameters()), "D": sum(p.numel() for p in net_D.parameters())},
           "init_checksum": {"G": checksum(torch.cat([p.flatten() for p in net_G.parameters()])),
                             "D": checksum(torch.cat([p.flatten() for p in net_D.parameters()]))}}
    gen = torch.Generator().manual_seed(SEED + 4)
    real = torch.rand(B, 3, 128, 128, generator=gen) * 2 - 1
    z1 = torch.rand(B, 120, generator=gen); z2 = torch.rand(B, 120, generator=gen)
    labels = torch.randint(0, ncls - 1, (3, B), generator=gen)
    rec["input_checksum"] = {"real": checksum(real), "z1": checksum(z1), "labels": labels.tolist()}
    blob = {}
    req = lambda m, f: [p.requires_grad_(f) for p in m.parameters()]
    req(net_G, False); req(net_D, True)
    D_x, D_G_z1 = losses.get_hinge_loss(net_G, net_D, real, labels[0], z1, labels[1])
    orth_reg(net_D)
    blob["d.scalars"] = np.array([D_x.item(), D_G_z1.item()])
    for k, p in net_D.named_parameters():
        blob["d.gradD." + k] = _sample(p.grad.numpy())
        blob["d.gradD_norm." + k] = np.array([np.linalg.norm(p.grad.numpy().astype(np.float64))])
        if wide:
            blob["d.gradD_proj." + k] = _projections(p.grad)
    req(net_G, True); req(net_D, False)
    G_error = losses.get_hinge_loss(net_G, net_D, noise_vector=z2, fake_label=labels[2], discriminator=False)
    orth_reg(net_G)
    blob["g.scalars"] = np.array([G_error.item()])
    for k, p in net_G.named_parameters():
        blob["g.gradG." + k] = _sample(p.grad.numpy())
        blob["g.gradG_norm." + k] = np.array([np.linalg.norm(p.grad.numpy().astype(np.float64))])
        if wide:
            blob["g.gradG_proj." + k] = _projections(p.grad)
    with torch.no_grad():
        img = net_G(z1, labels[1])
    blob["image_crop"] = img[:, :, 32:96, 32:96].numpy()
    for k, v in net_G.state_dict().items():
        if "U_vector" in k and v.numel() <= 64:
            blob["after.G." + k] = v.clone().numpy()
    np.savez_compressed(os.path.join(out, name + ".npz"), **blob)
    json.dump(rec, open(os.path.join(out, name + ".json"), "w"), indent=1)
    print(name + ".npz:", len(blob), "arrays;", blob["d.scalars"], blob["g.scalars"], rec["param_count"])


# --------------------------------------------------------------------------------------- loss curves
def synth_inputs(gen, B, nz, res, n_z):
    """Synthetic inputs drawn from a dedicated CPU generator so that the same numbers can be regenerated
    on any machine with the same torch version (checksums are stored to detect drift)."""
    real = torch.rand(B, 3, res, res, generator=gen) * 2 - 1
    zs = [torch.randn(B, nz, 1, 1, generator=gen) for _ in range(n_z)]
    return real, zs


def make_dcgan_losses(ref, out, steps=3):
    models = load_by_path("ref_basic_models", os.path.join(ref, "BasicGANs/models.py"))
    losses = load_by_path("ref_basic_losses", os.path.join(ref, "BasicGANs/loss_functions.py"))
    seed_all()
    B, hd, nz, res = 128, 64, 100, 64
    net_G = models.Generator(nz, hd, 3, resolution=res); net_G.apply(weights_init)
    net_D = models.Discriminator(3, hd, resolution=res); net_D.apply(weights_init)
    rec = {"config": {"B": B, "hidden_dim": hd, "nz": nz, "res": res, "seed": SEED, "lr": 2e-4, "betas": [0.5, 0.999],
                      "data_seed": SEED + 1},
           "param_count": {"G": sum(p.numel() for p in net_G.parameters()), "D": sum(p.numel() for p in net_D.parameters())},
           "init_checksum": {"G": checksum(torch.cat([p.flatten() for p in net_G.parameters()])),
                             "D": checksum(torch.cat([p.flatten() for p in net_D.parameters()]))},
           "steps": []}
    opt_D = torch.optim.Adam(net_D.parameters(), lr=2e-4, betas=(0.5, 0.999))   # BasicGANs/DCGAN.py:156-157
    opt_G = torch.optim.Adam(net_G.parameters(), lr=2e-4, betas=(0.5, 0.999))
    criterion = torch.nn.BCELoss()
    gen = torch.Generator().manual_seed(SEED + 1)
    for it in range(steps):
        real, (z,) = synth_inputs(gen, B, nz, res, 1)
        if it == 0:
            rec["input_checksum"] = {"real": checksum(real), "z": checksum(z)}
        net_D.zero_grad()
        label = torch.full((B,), 1.0)
        D_error, D_G_z1, D_x, fake = losses.get_minimax_loss(net_G, net_D, criterion, label, real, z, 0.0)
        opt_D.step()
        net_G.zero_grad()
        label.fill_(1.0)
        G_error, D_G_z2 = losses.get_minimax_loss(net_G, net_D, criterion, label, fake, discriminator=False)
        opt_G.step()
        rec["steps"].append({"D_error": D_error.item(), "D_G_z1": D_G_z1, "D_x": D_x, "G_error": G_error.item(), "D_G_z2": D_G_z2})
        print("dcgan step", it, rec["steps"][-1])
    json.dump(rec, open(os.path.join(out, "dcgan_losses.json"), "w"), indent=1)


def make_wgangp_losses(ref, out, cycles=1, B=64, name="wgangp_losses"):
    models = load_by_path("ref_basic_models", os.path.join(ref, "BasicGANs/models.py"))
    losses = load_by_path("ref_basic_losses", os.path.join(ref, "BasicGANs/loss_functions.py"))
    seed_all()
    hd, nz, res = 64, 100, 64
    net_G = models.Generator(nz, hd, 3, resolution=res); net_G.apply(weights_init)
    net_D = models.Discriminator(3, hd, resolution=res, is_wgan=True); net_D.apply(weights_init)
    rec = {"config": {"B": B, "hidden_dim": hd, "nz": nz, "res": res, "seed": SEED, "lr": 1e-4, "betas": [0.0, 0.9],
                      "c_lambda": 10, "disc_repeats": 5, "data_seed": SEED + 1},
           "param_count": {"G": sum(p.numel() for p in net_G.parameters()), "D": sum(p.numel() for p in net_D.parameters())},
           "init_checksum": {"G": checksum(torch.cat([p.flatten() for p in net_G.parameters()])),
                             "D": checksum(torch.cat([p.flatten() for p in net_D.parameters()]))},
           "cycles": []}
    opt_D = torch.optim.Adam(net_D.parameters(), lr=1e-4, betas=(0.0, 0.9))     # BasicGANs/train.py:79-80
    opt_G = torch.optim.Adam(net_G.parameters(), lr=1e-4, betas=(0.0, 0.9))
    gen = torch.Generator().manual_seed(SEED + 1)
    orig_rand = torch.rand
    for c in range(cycles):
        real, zs = synth_inputs(gen, B, nz, res, 6)
        eps_all = orig_rand(5, B, 1, 1, 1, generator=gen)
        if c == 0:
            rec["input_checksum"] = {"real": checksum(real), "z0": checksum(zs[0]), "eps": checksum(eps_all)}
        crit = []
        for it in range(5):                                                      # BasicGANs/train.py:95-107
            opt_D.zero_grad()
            # the loss draws epsilon with torch.rand(..., requires_grad=True); substitute the pre-drawn value
            torch.rand = lambda *a, **kw: eps_all[it].clone().requires_grad_(kw.get("requires_grad", False))
            try:
                D_x, D_G_z1, gp = losses.get_wasserstein_loss(net_G, net_D, real, zs[it])
            finally:
                torch.rand = orig_rand
            opt_D.step()
            crit.append({"D_x": D_x.item(), "D_G_z1": D_G_z1, "gp": gp.item()})
            print("wgangp critic", c, it, crit[-1])
        opt_G.zero_grad()
        G_error, D_G_z2 = losses.get_wasserstein_loss(net_G, net_D, noise_vector=zs[5], discriminator=False, c_lambda=10)
        opt_G.step()
        rec["cycles"].append({"critic": crit, "G_error": G_error.item(), "D_G_z2": D_G_z2})
        print("wgangp gen", c, rec["cycles"][-1]["G_error"])
    json.dump(rec, open(os.path.join(out, name + ".json"), "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    todo = a.only.split(",") if a.only else ["ops", "wide_ops", "dcgan_trace", "wgangp_trace", "dcgan_losses", "wgangp_losses", "pix2pix_trace", "pix2pixhd_trace", "stylegan_trace", "stylegan256_trace", "biggan_trace",
                                                   "pix2pix_wide_trace", "biggan_wide_trace", "wgangp_losses_b256"]
    if "ops" in todo: make_ops(a.out)
    if "wide_ops" in todo: make_wide_ops(a.out)
    if "dcgan_trace" in todo: make_dcgan_trace(a.ref, a.out)
    if "wgangp_trace" in todo: make_wgangp_trace(a.ref, a.out)
    if "dcgan_losses" in todo: make_dcgan_losses(a.ref, a.out)
    if "wgangp_losses" in todo: make_wgangp_losses(a.ref, a.out)
    if "pix2pix_trace" in todo: make_pix2pix_trace(a.ref, a.out)
    if "pix2pixhd_trace" in todo: make_pix2pixhd_trace(a.ref, a.out)
    if "stylegan_trace" in todo: make_stylegan_trace(a.ref, a.out)
    if "stylegan256_trace" in todo: make_stylegan_trace(a.ref, a.out, pg=6, name="stylegan256_trace")
    if "biggan_trace" in todo: make_biggan_trace(a.ref, a.out)
    # the reference's full widths, whole models (batch 1 / 2) and the benchmark's batch for the WGAN-GP loss sequence
    if "pix2pix_wide_trace" in todo: make_pix2pix_trace(a.ref, a.out, B=1, ch=64, name="pix2pix_wide_trace")
    if "biggan_wide_trace" in todo: make_biggan_trace(a.ref, a.out, B=2, ch=64, name="biggan_wide_trace")
    if "wgangp_losses_b256" in todo: make_wgangp_losses(a.ref, a.out, B=256, name="wgangp_losses_b256")
