"""Pins the CPU oracle (oracle/conv_oracle.c, oracle/ref_models.py) against the golden vectors captured from
the reference's own files (tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ref_models as R


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def ops(golden_dir):
    return np.load(os.path.join(golden_dir, "ops.npz"))


def op_names(golden_dir=os.path.join(os.path.dirname(__file__), "golden")):
    z = np.load(os.path.join(golden_dir, "ops.npz"))
    return sorted({k.rsplit(".", 1)[0] for k in z.files if k.endswith(".meta")})


@pytest.mark.parametrize("name", op_names())
def test_c_oracle_matches_reference_ops(ops, name):
    N, Cin, H, W, Cout, k, s, p, has_b, op, is_t = [int(v) for v in ops[name + ".meta"]]
    x, w, y, gy, gx, gw = (ops[f"{name}.{t}"] for t in ("x", "w", "y", "gy", "gx", "gw"))
    b = ops[name + ".b"] if has_b else None
    tol = 2e-6  # oneDNN fp32 accumulation vs double accumulation in the oracle
    if not is_t:
        assert rel_err(oracle.conv2d_fprop(x, w, b, s, p), y) < tol
        assert rel_err(oracle.conv2d_dgrad(gy, w, x.shape, s, p), gx) < tol
        got_gw, got_gb = oracle.conv2d_wgrad(x, gy, w.shape, s, p, with_bias=True)
        assert rel_err(got_gw, gw) < tol
        if has_b:
            assert rel_err(got_gb, ops[name + ".gb"]) < tol
    else:
        assert rel_err(oracle.conv_transpose2d_fprop(x, w, b, s, p, op), y) < tol
        # a transposed conv is the dgrad of the conv whose OIHW weight is this IOHW weight, so its own
        # backward passes are that conv's fprop (dx) and wgrad with the roles of x and gy swapped (dw)
        assert rel_err(oracle.conv2d_dgrad(x, w, y.shape, s, p) + (0 if b is None else b[None, :, None, None]), y) < tol
        assert rel_err(oracle.conv2d_fprop(gy, w, None, s, p), gx) < tol
        got_gw, got_gb = oracle.conv2d_wgrad(gy, x, w.shape, s, p, with_bias=False), gy.sum((0, 2, 3))
        assert rel_err(got_gw, gw) < tol
        if has_b:
            assert rel_err(got_gb, ops[name + ".gb"]) < 1e-5


def test_c_oracle_blur(ops):
    assert rel_err(oracle.dwblur3x3(ops["blur.x"]), ops["blur.y"]) < 1e-6


def test_round_bf16_matches_torch():
    x = torch.randn(10000) * 3
    x[0] = float("nan"); x[1] = float("inf"); x[2] = 0.0; x[3] = 1e-40
    got = oracle.round_bf16(x.numpy())
    exp = x.bfloat16().float().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    m = ~np.isnan(exp)
    assert np.array_equal(got[m], exp[m])


def _load_state(net, blob, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(blob[k]) for k in blob.files if k.startswith(prefix)}
    net.load_state_dict(sd)


def test_restated_dcgan_step_matches_reference_trace(golden_dir):
    z = np.load(os.path.join(golden_dir, "dcgan_trace.npz"))
    B, hd = z["real"].shape[0], z["init.D.main.0.0.weight"].shape[0]
    G = R.Generator(100, hd, 3, resolution=64); D = R.Discriminator(3, hd, resolution=64)
    _load_state(G, z, "init.G."); _load_state(D, z, "init.D.")
    real, noise = torch.from_numpy(z["real"]), torch.from_numpy(z["z"])
    crit = torch.nn.BCELoss()
    D.zero_grad()
    label = torch.full((B,), 1.0)
    d_err, d_g_z1, d_x, fake = R.minimax_step(G, D, crit, label, real, noise, 0.0)
    np.testing.assert_allclose([d_err.item(), d_g_z1, d_x], z["dphase.scalars"], rtol=1e-6)
    for k, p in D.named_parameters():
        assert rel_err(p.grad.numpy(), z["dphase.gradD." + k]) < 1e-6, k
    G.zero_grad(); label.fill_(1.0)
    g_err, d_g_z2 = R.minimax_step(G, D, crit, label, fake, discriminator=False)
    np.testing.assert_allclose([g_err.item(), d_g_z2], z["gphase.scalars"], rtol=1e-6)
    for k, p in G.named_parameters():
        assert rel_err(p.grad.numpy(), z["gphase.gradG." + k]) < 1e-6, k
    for k, v in G.state_dict().items():
        if "running" in k:
            assert rel_err(v.numpy(), z["after.G." + k]) < 1e-6, k


def test_c_oracle_on_reference_layer_traces(golden_dir):
    """Every conv layer call recorded inside the reference's DCGAN step: C oracle reproduces y and gx."""
    z = np.load(os.path.join(golden_dir, "dcgan_trace.npz"))
    calls = sorted({k.rsplit(".", 1)[0] for k in z.files if k.startswith("trace.")})
    assert len(calls) >= 20
    for c in calls:
        net, layer = c.split(".")[1], ".".join(c.split(".")[3:])
        w = z[f"init.{net}.{layer}.weight"]
        x, y = z[c + ".x"], z[c + ".y"]
        first, last = layer == "main.0.0", layer == "main.4.0"
        if net == "D":
            s, p = (1, 0) if last else (2, 1)
            assert rel_err(oracle.conv2d_fprop(x, w, None, s, p), y) < 2e-6, c
            if c + ".gx" in z.files:
                assert rel_err(oracle.conv2d_dgrad(z[c + ".gy"], w, x.shape, s, p), z[c + ".gx"]) < 2e-6, c
        else:
            s, p = (1, 0) if first else (2, 1)
            assert rel_err(oracle.conv_transpose2d_fprop(x, w, None, s, p), y) < 2e-6, c
            if c + ".gx" in z.files:
                assert rel_err(oracle.conv2d_fprop(z[c + ".gy"], w, None, s, p), z[c + ".gx"]) < 2e-6, c


def test_restated_wgangp_matches_reference_trace(golden_dir):
    z = np.load(os.path.join(golden_dir, "wgangp_trace.npz"))
    hd = z["init.D.main.0.0.weight"].shape[0]
    G = R.Generator(100, hd, 3, resolution=64); D = R.Discriminator(3, hd, resolution=64, is_wgan=True)
    _load_state(G, z, "init.G."); _load_state(D, z, "init.D.")
    real, noise, noise1 = (torch.from_numpy(z[k]) for k in ("real", "z", "z1"))
    d_x, d_g_z1, gp = R.wasserstein_step(G, D, real, noise, epsilon=torch.from_numpy(z["epsilon"]))
    np.testing.assert_allclose([d_x.item(), d_g_z1, gp.item()], z["critic.scalars"], rtol=2e-5)
    for k, p in D.named_parameters():
        assert rel_err(p.grad.numpy(), z["critic.gradD." + k]) < 2e-5, k
    G.zero_grad(); D.zero_grad()
    g_err, d_g_z2 = R.wasserstein_step(G, D, noise_vector=noise1, discriminator=False)
    np.testing.assert_allclose([g_err.item(), d_g_z2], z["gen.scalars"], rtol=2e-5)
    for k, p in G.named_parameters():
        assert rel_err(p.grad.numpy(), z["gen.gradG." + k]) < 2e-5, k


def test_restated_dcgan_full_size_losses(golden_dir):
    """Full-size DCGAN-64 (hidden 64, batch 128): first step's losses with regenerated inputs and weights."""
    rec = json.load(open(os.path.join(golden_dir, "dcgan_losses.json")))
    cfg = rec["config"]
    torch.manual_seed(cfg["seed"])
    G = R.Generator(cfg["nz"], cfg["hidden_dim"], 3, resolution=cfg["res"]); G.apply(R.weights_init)
    D = R.Discriminator(3, cfg["hidden_dim"], resolution=cfg["res"]); D.apply(R.weights_init)
    assert sum(p.numel() for p in G.parameters()) == rec["param_count"]["G"] == 3576704
    assert sum(p.numel() for p in D.parameters()) == rec["param_count"]["D"] == 2765568
    opt_D = torch.optim.Adam(D.parameters(), lr=cfg["lr"], betas=tuple(cfg["betas"]))
    opt_G = torch.optim.Adam(G.parameters(), lr=cfg["lr"], betas=tuple(cfg["betas"]))
    gen = torch.Generator().manual_seed(cfg["data_seed"])
    real, (noise,) = R.synth_inputs(gen, cfg["B"], cfg["nz"], cfg["res"], 1)
    out = R.dcgan_train_step(G, D, opt_G, opt_D, real, noise)
    for k, v in rec["steps"][0].items():
        assert abs(out[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, out[k], v)


def _checksum(t):
    import hashlib
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]


def pix2pix_setup(golden_dir, make_G, make_D, name="pix2pix_trace"):
    """Nets re-initialised from the reference's seed (PyTorch default init, as the reference applies none),
    inputs regenerated from the data seed, dropout masks from the fixture (`pix2pix_wide_trace`, the reference's full
    width at batch 1, keeps them as one bit per element)."""
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rec = json.load(open(os.path.join(golden_dir, name + ".json")))
    B, ch = [int(v) for v in z["cfg"]]
    torch.manual_seed(5000)
    G = make_G(ch); D = make_D(ch)
    gen = torch.Generator().manual_seed(5002)
    real = torch.rand(B, 3, 256, 256, generator=gen) * 2 - 1
    label = torch.rand(B, 3, 256, 256, generator=gen)
    assert _checksum(real) == rec["input_checksum"]["real"] and _checksum(label) == rec["input_checksum"]["label"]
    n_d = int(z["n_d_masks"][0])
    def mask(i):
        m = z[f"mask{i}"]
        if m.dtype == np.uint8:        # packed bits: kept elements carry 1 / (1 - p) = 2
            shape = tuple(int(v) for v in z[f"mask{i}.shape"])
            m = np.unpackbits(m)[:int(np.prod(shape))].reshape(shape).astype(np.float32) * 2.0
        return torch.from_numpy(m)
    masks = [mask(i) for i in range(2 * n_d)]
    return z, rec, G, D, real, label, masks[:n_d], masks[n_d:]


def pix2pixhd_setup(golden_dir, make_G):
    """Generator re-initialised from the reference's seed (PyTorch default init), inputs and dropout masks from the fixture."""
    z = np.load(os.path.join(golden_dir, "pix2pixhd_trace.npz"))
    rec = json.load(open(os.path.join(golden_dir, "pix2pixhd_trace.json")))
    B, ch, blocks, res = [int(v) for v in z["cfg"]]
    torch.manual_seed(5000)
    G = make_G(ch, blocks)
    assert sum(p.numel() for p in G.parameters()) == rec["param_count"]
    assert _checksum(torch.cat([p.flatten() for p in G.parameters()])) == rec["init_checksum"]
    assert list(G.state_dict().keys()) == rec["state_dict_keys"]
    masks = [torch.from_numpy(z[f"mask{i}"]) for i in range(blocks)]
    return z, G, torch.from_numpy(z["label"]), torch.from_numpy(z["target"]), masks


def test_restated_pix2pixhd_generator_matches_reference(golden_dir):
    """Pix2Pix/models.py:150-218 (GeneratorHD, ResNetBlock): output, loss, input gradient, every parameter gradient and
    the BatchNorm running statistics of one forward + backward."""
    z, G, label, target, masks = pix2pixhd_setup(golden_dir, lambda ch, nb: R.P2PGeneratorHD(ch=ch, n_blocks=nb))
    label.requires_grad_(True)
    fake = G(label, masks)
    loss = torch.nn.functional.l1_loss(fake, target)
    loss.backward()
    assert abs(loss.item() - z["loss"][0]) < 1e-6
    assert rel_err(fake.detach().numpy(), z["fake"]) < 1e-5
    assert rel_err(label.grad.numpy(), z["grad_label"]) < 1e-4
    for k, p in G.named_parameters():
        assert rel_err(p.grad.numpy(), z["grad." + k]) < 1e-4 or np.linalg.norm(z["grad." + k]) < 1e-7, k
    for k, b in G.named_buffers():
        if "running" in k:
            assert rel_err(b.numpy(), z["buf." + k]) < 1e-5, k


@pytest.mark.parametrize("fixture", ["pix2pix_trace", "pix2pix_wide_trace"], ids=["ch8_b2", "ch64_b1"])
def test_restated_pix2pix_step_matches_reference(golden_dir, fixture):
    z, rec, G, D, real, label, masks_d, masks_g = pix2pix_setup(golden_dir, lambda ch: R.P2PGenerator(ch=ch), lambda ch: R.P2PDiscriminator(ch=ch), fixture)
    assert sum(p.numel() for p in G.parameters()) == rec["param_count"]["G"]
    assert _checksum(torch.cat([p.flatten() for p in G.parameters()])) == rec["init_checksum"]["G"]
    assert _checksum(torch.cat([p.flatten() for p in D.parameters()])) == rec["init_checksum"]["D"]
    d_loss = R.pix2pix_discriminator_phase(G, D, real, label, masks_d)
    assert abs(d_loss.item() - z["d.loss"][0]) < 1e-6
    if fixture == "pix2pix_wide_trace":
        assert check_grads(D.named_parameters(), z, "d.gradD", 2e-5) >= 10
    else:
        biggest = max(np.linalg.norm(z["d.gradD." + k]) for k, _ in D.named_parameters())
        for k, p in D.named_parameters():
            want = z["d.gradD." + k]
            if np.linalg.norm(want) < 1e-5 * biggest:          # a bias in front of BatchNorm: rounding residue only
                assert np.linalg.norm(p.grad.numpy()) < 1e-5 * biggest, k
            else:
                assert rel_err(p.grad.numpy(), want) < 2e-5, k
    g_err, fake = R.pix2pix_generator_phase(G, D, real, label, 100.0, masks_g)
    assert abs(g_err.item() - z["g.loss"][0]) < 1e-5 * z["g.loss"][0]
    assert rel_err(fake[:, :, 96:160, 96:160].numpy(), z["g.fake_crop"]) < 1e-5
    biggest = max(z["g.gradG_norm." + k][0] for k, _ in G.named_parameters())
    for k, p in G.named_parameters():
        got, want = np.linalg.norm(p.grad.numpy().astype(np.float64)), z["g.gradG_norm." + k][0]
        if want < 1e-5 * biggest:                              # a bias in front of BatchNorm: rounding residue only
            assert got < 1e-5 * biggest, k
        else:
            assert abs(got - want) < 1e-4 * want, k


def test_full_size_pix2pix_parameter_counts():
    G, D = R.P2PGenerator(), R.P2PDiscriminator()
    assert sum(p.numel() for p in G.parameters()) == 54419459
    assert sum(p.numel() for p in D.parameters()) == 2769601
    with torch.no_grad():
        assert D(torch.zeros(1, 3, 256, 256), torch.zeros(1, 3, 256, 256)).shape == (1, 1, 30, 30)


def stylegan_setup(golden_dir, make_G, make_D, name="stylegan_trace"):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rec = json.load(open(os.path.join(golden_dir, name + ".json")))
    torch.manual_seed(5000)
    G = make_G(); D = make_D()
    gen = torch.Generator().manual_seed(5003)
    res = 4 * 2 ** rec["pg_step"]
    real = torch.rand(rec["B"], 3, res, res, generator=gen) * 2 - 1
    zs = torch.randn(4, rec["B"], 512, generator=gen)
    assert _checksum(real) == rec["input_checksum"]["real"] and _checksum(zs) == rec["input_checksum"]["zs"]
    noises = {ph: [torch.from_numpy(z[f"{ph}.noise{i}"]) for i in range(rec["pg_step"] + 1)] for ph in ("critic", "gen", "fade")}
    return z, rec, G, D, real, zs, noises


def projections(t, seed=29, n=8):
    """make_golden._projections: n random +-1 projections of the whole tensor in float64."""
    flat = t.detach().reshape(-1).double().cpu()
    g = torch.Generator().manual_seed(seed)
    return np.array([float((flat * (torch.randint(0, 2, (flat.numel(),), generator=g, dtype=torch.int8).double() * 2 - 1)).sum())
                     for _ in range(n)])


def check_grads(named_params, z, prefix, tol, report=None, tol_of=None):
    """Sampled gradient entries, full norms and (where the fixture holds them) +-1 projections of the whole tensor
    recorded by make_golden.  tol_of(name, parameter) overrides `tol` per parameter."""
    biggest = max(z[k][0] for k in z.files if k.startswith(prefix + "_norm."))
    n = 0
    for k, p in named_params:
        key = prefix + "." + k
        if key not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        g = p.grad.detach().cpu().numpy()
        want_norm = z[prefix + "_norm." + k][0]
        if want_norm < 1e-6 * biggest:
            # analytically zero gradients (a bias in front of a normalisation): rounding residue only, which scales
            # with the comparison's tolerance (bf16-stored activations leave about 2e-4 of the largest gradient)
            assert np.linalg.norm(g) < max(1e-4, 2e-3 * tol) * biggest, k
            continue
        gs = g if g.size <= 20000 else g.ravel()[::max(1, g.size // 4096)][:4096]
        err = rel_err(gs, z[key])
        if report is not None:
            report[k] = err
        lim = tol if tol_of is None else tol_of(k, p)
        # A gradient whose norm is r times the largest one carries the rounding residue of the whole backward pass
        # (fp32, whose order of additions differs between host CPU types) at about 1e-5 / r of its own size.
        lim = max(lim, 1e-5 * biggest / want_norm)
        assert err < lim, (k, err, lim)
        assert abs(np.linalg.norm(g.astype(np.float64)) - want_norm) < lim * want_norm, k
        pkey = prefix + "_proj." + k
        if pkey in z.files:          # a projection is a sum over every element: the error of a random +-1 sum of N
            want = z[pkey]           # elements with relative error e is about e * |g|, whatever the projection's own size
            assert np.abs(projections(p.grad) - want).max() < 4 * lim * want_norm, (k, projections(p.grad), want, want_norm)
        n += 1
    return n


@pytest.mark.parametrize("fixture", ["stylegan_trace", "stylegan256_trace"], ids=["128x128_step5", "256x256_step6"])
def test_restated_stylegan_matches_reference(golden_dir, fixture):
    from oracle import ref_stylegan as S
    z, rec, G, D, real, zs, noises = stylegan_setup(golden_dir, lambda: S.StyleBasedGenerator(fused=True),
                                                    lambda: S.Discriminator(fused=True, use_activations=True), fixture)
    lo, hi = 2 * 2 ** rec["pg_step"] - 32, 2 * 2 ** rec["pg_step"] + 32          # the recorded image window
    assert sum(p.numel() for p in G.parameters()) == rec["param_count"]["G"] == 26219115
    assert sum(p.numel() for p in D.parameters()) == rec["param_count"]["D"] == 26054017
    assert _checksum(torch.cat([p.flatten() for p in G.parameters()])) == rec["init_checksum"]["G"]
    assert _checksum(torch.cat([p.flatten() for p in D.parameters()])) == rec["init_checksum"]["D"]
    pg = rec["pg_step"]
    d_x, d_g, gp, _ = S.critic_phase(G, D, real, [zs[0], zs[1]], torch.from_numpy(z["critic.eps"]), pg, 1, 10.0,
                                     noises["critic"], int(z["critic.crossover"][0]))
    np.testing.assert_allclose([d_x.item(), d_g.item(), gp.item()], z["critic.scalars"], rtol=2e-4)
    assert check_grads(D.named_parameters(), z, "critic.gradD", 2e-3) > 20
    g_err, _ = S.generator_phase(G, D, [zs[2], zs[3]], pg, 1, noises["gen"], int(z["gen.crossover"][0]))
    np.testing.assert_allclose(g_err.item(), z["gen.scalars"][0], rtol=2e-4)
    # a noise layer's scale sums gradient x noise over every pixel (a cancelling sum of 1e5..2e6 terms per channel):
    # between host CPU types its fp32 value moves by up to 1e-2 of itself
    assert check_grads(G.named_parameters(), z, "gen.gradG", 5e-3,
                       tol_of=lambda k, p: 2e-2 if "noise_injection" in k else 5e-3) > 40
    with torch.no_grad():
        img = G(zs[0], progressive_growing_phase=pg, alpha=0.5, noises=noises["fade"])
        score = D(img, progressive_growing_phase=pg, alpha=0.5)
    assert rel_err(img[:, :, lo:hi, lo:hi].numpy(), z["fade.image_crop"]) < 1e-5
    assert rel_err(score.numpy(), z["fade.score"]) < 1e-4


def biggan_setup(golden_dir, M, name="biggan_trace"):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    rec = json.load(open(os.path.join(golden_dir, name + ".json")))
    torch.manual_seed(5000)
    G = M.Generator(n_classes=rec["n_classes"], ch=rec["ch"]); D = M.Discriminator(n_classes=rec["n_classes"], ch=rec["ch"])
    M.orthogonal_init(G); M.orthogonal_init(D)
    gen = torch.Generator().manual_seed(5004)
    B = rec["B"]
    real = torch.rand(B, 3, 128, 128, generator=gen) * 2 - 1
    z1 = torch.rand(B, 120, generator=gen); z2 = torch.rand(B, 120, generator=gen)
    labels = torch.randint(0, rec["n_classes"] - 1, (3, B), generator=gen)
    assert _checksum(real) == rec["input_checksum"]["real"] and labels.tolist() == rec["input_checksum"]["labels"]
    return z, rec, G, D, real, z1, z2, labels


@pytest.mark.parametrize("fixture", ["biggan_trace", "biggan_wide_trace"], ids=["ch8_b4", "ch64_b2"])
def test_restated_biggan_matches_reference(golden_dir, fixture):
    from oracle import ref_biggan as Bg
    z, rec, G, D, real, z1, z2, labels = biggan_setup(golden_dir, Bg, fixture)
    # No bit-for-bit checksum of the initial weights here: orthogonal initialisation runs a QR factorisation in LAPACK,
    # whose last bits depend on the host CPU type.  The losses, gradients and updated weights below pin it to 1e-5.
    assert sum(p.numel() for p in G.parameters()) == rec["param_count"]["G"]
    assert sum(p.numel() for p in D.parameters()) == rec["param_count"]["D"]
    d_x, d_g = Bg.discriminator_phase(G, D, real, labels[0], z1, labels[1])
    np.testing.assert_allclose([d_x.item(), d_g.item()], z["d.scalars"], rtol=1e-5, atol=1e-7)
    assert check_grads(D.named_parameters(), z, "d.gradD", 1e-3) > 20
    g_err = Bg.generator_phase(G, D, z2, labels[2])
    np.testing.assert_allclose(g_err.item(), z["g.scalars"][0], rtol=1e-5)
    assert check_grads(G.named_parameters(), z, "g.gradG", 1e-3) > 40
    with torch.no_grad():
        img = G(z1, labels[1])
    assert rel_err(img[:, :, 32:96, 32:96].numpy(), z["image_crop"]) < 1e-5
    for k, v in G.state_dict().items():
        if "after.G." + k in z.files:
            assert rel_err(v.numpy(), z["after.G." + k]) < 1e-5, k


def test_full_size_biggan_parameter_counts():
    from oracle import ref_biggan as Bg
    assert sum(p.numel() for p in Bg.Generator(n_classes=4, ch=64).parameters()) == 31859620
    assert sum(p.numel() for p in Bg.Discriminator(n_classes=4, ch=64).parameters()) == 39478034


def test_c_oracle_on_full_width_layer_digests(golden_dir):
    """The C restatement of the three passes on layers at the reference's full channel widths (the Pix2Pix U-Net bottom
    and a BigGAN 1024 -> 1024 block; the larger spatial cases are left to the GPU test, the scalar C loops would take
    minutes), against the digests of the reference's torch calls in wide_ops.npz."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_dir, "make_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    z = np.load(os.path.join(golden_dir, "wide_ops.npz"))
    for case in mk.WIDE_CASES:
        name, kind, N, Cin, H, W, Cout, k, s, p = case
        if not (name.startswith("p2p_") or name == "biggan_1024_8x8"):
            continue
        x, w, gy = (t.numpy() for t in mk.wide_inputs(case))
        if kind == "conv":
            got = {"y": oracle.conv2d_fprop(x, w, None, s, p), "gx": oracle.conv2d_dgrad(gy, w, x.shape, s, p),
                   "gw": oracle.conv2d_wgrad(x, gy, w.shape, s, p)}
        else:
            got = {"y": oracle.conv2d_dgrad(x, w, gy.shape, s, p), "gx": oracle.conv2d_fprop(gy, w, None, s, p),
                   "gw": oracle.conv2d_wgrad(gy, x, w.shape, s, p)}
        for tag, t in got.items():
            sample, proj, norm = mk.wide_digest(torch.from_numpy(np.ascontiguousarray(t)), 17)
            want_n = float(z[f"{name}.{tag}.norm"])
            err = np.linalg.norm(sample - z[f"{name}.{tag}.sample"]) / np.linalg.norm(z[f"{name}.{tag}.sample"])
            assert err < 2e-5, (name, tag, err)
            assert np.all(np.abs(proj - z[f"{name}.{tag}.proj"]) < 1e-4 * want_n), (name, tag)
