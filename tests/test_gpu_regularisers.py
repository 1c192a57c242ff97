"""BigGAN's weight-side operations on library kernels: the power iteration of SpectralNorm (BigGAN/model.py:20-35) and
the orthogonal-regularisation gradient (BigGAN/train.py:115-127), against the reference's own expressions evaluated in
float64 on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    return gan_amd


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def _power_iterations(w, u, n):           # BigGAN/model.py:27-35 in float64
    w = w.double().flatten(1)
    u = u.double().reshape(1, -1)
    for _ in range(n):
        v = torch.nn.functional.normalize(u @ w, p=2)
        u = torch.nn.functional.normalize(v @ w.t(), p=2)
        s = (u @ w @ v.t()).squeeze()
    return s, u


@pytest.mark.parametrize("shape", [(64, 32, 3, 3), (128, 3, 3, 3), (96, 20), (24576, 20), (1024, 1024, 3, 3), (8, 8, 1, 1), (1, 37)])
@pytest.mark.parametrize("n_iterations", [1, 3])
def test_spectral_norm_power_iteration(gan, shape, n_iterations):
    g = torch.Generator().manual_seed(51)
    w = torch.randn(*shape, generator=g)
    u0 = torch.rand(1, shape[0], generator=g)
    want_s, want_u = _power_iterations(w, u0, n_iterations)
    u = u0.clone().cuda()
    s = gan.ops.spectral_sigma_(w.cuda(), u, n_iterations)
    assert s.shape == () and u.shape == (1, shape[0])
    assert abs(s.item() - want_s.item()) <= 2e-5 * abs(want_s.item())
    assert rel(u, want_u) < 2e-5


def test_spectral_norm_layer_matches_reference_layer(gan):
    """The harness's Conv2d against the reference's formula on the same weight and U_vector: normalised weight, advanced
    buffer, and a second call that starts from the advanced buffer."""
    from gan_amd.harness import biggan as B
    torch.manual_seed(52)
    layer = B.Conv2d(48, 64, 3, padding=1).cuda()
    u0 = layer.U_vector.detach().cpu().clone()
    x = torch.randn(2, 48, 8, 8, device="cuda")
    for _ in range(2):
        want_s, want_u = _power_iterations(layer.weight.detach().cpu(), u0, 1)
        y = layer(x)
        ref = torch.nn.functional.conv2d(x.cpu().double(), layer.weight.detach().cpu().double() / want_s, layer.bias.detach().cpu().double(), padding=1)
        assert rel(layer.U_vector, want_u) < 2e-5
        assert rel(y, ref) < 6e-3           # bf16 operands
        u0 = want_u.float()


@pytest.mark.parametrize("shape", [(64, 32, 3, 3), (128, 3, 3, 3), (256, 256, 1, 1), (512, 512, 3, 3), (40, 24, 3, 3)])
def test_orthogonal_regularisation_gradient(gan, shape):
    g = torch.Generator().manual_seed(53)
    w = torch.randn(*shape, generator=g) * 0.05
    grad0 = torch.randn(*shape, generator=g) * 1e-3
    beta = 1e-4
    w2 = w.double().flatten(1)
    want = grad0.double() + beta * (2 * ((w2 @ w2.t()) * (1.0 - torch.eye(shape[0], dtype=torch.float64))) @ w2).view(shape)
    p = torch.nn.Parameter(w.cuda())
    p.grad = grad0.clone().cuda()
    skipped = torch.nn.Parameter(torch.randn(16, 16, generator=g).cuda())        # two dimensions: left alone
    skipped.grad = torch.ones(16, 16, device="cuda")
    gan.ops.orthogonal_regularization_([p, skipped], beta, precision="fp32x3")
    assert rel(p.grad - grad0.cuda(), want - grad0.double()) < 5e-5          # the added term itself
    assert torch.equal(skipped.grad, torch.ones(16, 16, device="cuda"))
    p.grad = grad0.clone().cuda()
    gan.ops.orthogonal_regularization_([p], beta)                            # the convolutions' mode: bf16 operands
    assert rel(p.grad - grad0.cuda(), want - grad0.double()) < 2e-2


def test_spectral_norms_of_a_whole_network_in_one_set_of_launches(gan):
    """ops.SpectralNormSet against the per-weight call: same sigmas, same advanced vectors, over two refreshes; and through the
    BigGAN harness, whose networks refresh every layer in front of a forward pass."""
    g = torch.Generator().manual_seed(54)
    shapes = [(64, 32, 3, 3), (128, 3, 3, 3), (96, 20), (2048, 20), (256, 256, 1, 1), (8, 8, 1, 1), (1, 37), (512, 512, 3, 3)]
    ws = [torch.randn(*s, generator=g).cuda() for s in shapes]
    u_one = [torch.rand(1, s[0], generator=g).cuda() for s in shapes]
    u_set = [u.clone() for u in u_one]
    st = gan.ops.SpectralNormSet(ws, u_set)
    for _ in range(2):
        sig = st.refresh().clone()
        for i, (w, u) in enumerate(zip(ws, u_one)):
            want = gan.ops.spectral_sigma_(w, u, 1)
            assert torch.equal(sig[i], want) and torch.equal(u_set[i], u), i
    from gan_amd.harness import biggan as B
    torch.manual_seed(55)
    D = B.Discriminator(n_classes=4, ch=8).cuda()
    layers = [m for m in D.modules() if isinstance(m, (B.Conv2d, B.Linear))]
    u0 = [m.U_vector.detach().cpu().clone() for m in layers]
    D(torch.randn(2, 3, 128, 128, device="cuda"), torch.zeros(2, dtype=torch.long, device="cuda"))
    for m, u in zip(layers, u0):
        _, want_u = _power_iterations(m.weight.detach().cpu(), u, 1)
        assert rel(m.U_vector, want_u) < 2e-5
        assert "_sigma_ahead" not in m.__dict__            # every layer took its sigma
