"""GPU tests of the modulated-convolution EXTENSION (gan_amd.functional.modulated_conv2d and the two kernels under it,
gc_channel_scale / gc_rowdot, through the C ABI).  Parity unpinned: the reference has no such layer; the checker is the
restatement of the published formula in oracle/ref_modconv.py (per-sample weights, grouped convolution) in float64."""
import numpy as np
import pytest
import torch

from oracle import ref_modconv

pytestmark = pytest.mark.gpu

TOL_X3, TOL_BF16 = 1e-4, 1e-2       # relative L2: three-term split (the fp32 bar) / bf16 operands


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gan_amd


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [(3, 8, 5, 7), (2, 64, 32, 32), (4, 12, 1, 1), (1, 16, 96, 96), (0, 8, 4, 4)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_channel_scale_and_rowdot_against_torch(gan, shape, dt):
    """Both kernels in both storage types: one slab per sample, several slabs (96 x 96), single pixels, an empty batch."""
    g = torch.Generator().manual_seed(3)
    N, C, H, W = shape
    x = torch.randn(shape, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    b = torch.randn(shape, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    s = torch.randn(N, C, generator=g).cuda()
    y = gan.ops.channel_scale(x, s)
    assert y.dtype == dt and y.shape == x.shape
    want = x.double() * s.double().view(N, C, 1, 1)
    if N:
        assert rel(y, want) < (1e-6 if dt == torch.float32 else 4e-3)
        if dt == torch.float32:
            assert torch.equal(y, x * s.view(N, C, 1, 1))                    # one rounded product per element
    r = gan.ops.rowdot(x, b)
    assert r.dtype == torch.float32 and r.shape == (N, C)
    if N:
        assert rel(r, (x.double() * b.double()).sum((2, 3))) < 2e-6
        assert torch.equal(r, gan.ops.rowdot(x, b))                          # fixed summation order


def test_channel_scale_pair_is_closed_under_differentiation(gan):
    """First and second derivatives of the pair against torch's autograd on the tensor expressions (float64, CPU)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 6, 5, generator=g)
    s = torch.randn(2, 8, generator=g)
    u = torch.randn(2, 8, 6, 5, generator=g)

    def run(xx, ss, uu, scale, dot):
        y = scale(xx, ss)
        gx, gs = torch.autograd.grad((y * uu).sum() + (y ** 2).sum(), [xx, ss], create_graph=True)
        return y, gx, gs, torch.autograd.grad(gx.square().sum() + gs.square().sum() + dot(gx, xx).sum(), [xx, ss])

    xr, sr = x.double().requires_grad_(True), s.double().requires_grad_(True)
    ref = run(xr, sr, u.double(), lambda a, b: a * b.view(2, 8, 1, 1), lambda a, b: (a * b).sum((2, 3)))
    xg, sg = x.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    got = run(xg, sg, u.cuda(), gan.ops.channel_scale, gan.ops.rowdot)
    for a, b, what in [(got[0], ref[0], "y"), (got[1], ref[1], "gx"), (got[2], ref[2], "gs"), (got[3][0], ref[3][0], "ggx"),
                       (got[3][1], ref[3][1], "ggs")]:
        assert rel(a, b) < 1e-5, what


CASES = [   # N, Cin, Cout, H, W, k, stride, padding, demodulate
    (3, 8, 12, 9, 7, 3, 1, 1, True),
    (2, 64, 64, 16, 16, 3, 1, 1, True),
    (2, 16, 8, 8, 8, 1, 1, 0, False),
    (4, 128, 64, 8, 8, 3, 1, 1, True),
    (2, 32, 32, 16, 16, 4, 2, 1, True),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("prec,tol", [("fp32x3", TOL_X3), ("bf16", TOL_BF16)])
def test_modulated_conv_against_oracle(gan, case, prec, tol):
    """Output and the gradients with respect to input, weight and style against the per-sample-weight formulation."""
    N, Cin, Cout, H, W, k, stride, pad, demod = case
    g = torch.Generator().manual_seed(11 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g)
    s = torch.randn(N, Cin, generator=g) * 0.5 + 1.0
    c = 1.0 / (Cin * k * k) ** 0.5
    xr, wr, sr = (t.double().requires_grad_(True) for t in (x, w, s))
    yr = ref_modconv.modulated_conv2d(xr, wr, sr, stride, pad, demod, 1e-8, c)
    seed = torch.randn(yr.shape, generator=g)
    gr = torch.autograd.grad(yr, [xr, wr, sr], seed.double())
    code = {"fp32x3": gan._lib.GC_PREC_F32X3, "bf16": gan._lib.GC_PREC_BF16}[prec]
    xg, wg, sg = (t.cuda().requires_grad_(True) for t in (x, w, s))
    yg = gan.functional.modulated_conv2d(xg, wg, sg, None, stride, pad, demod, 1e-8, weight_scale=c, precision=code)
    gg = torch.autograd.grad(yg, [xg, wg, sg], seed.cuda())
    assert rel(yg, yr) < tol, "output"
    for a, b, what in zip(gg, gr, ("input", "weight", "style")):
        assert rel(a, b) < tol, f"gradient by {what}"


def test_modulated_conv_second_order_against_oracle(gan):
    """A path-length-style penalty: the gradient of |d y / d style|^2 with respect to weight and style (double backward
    through both scalings, the convolution and the normalisation factor)."""
    N, Cin, Cout, H, W, k = 2, 8, 8, 6, 6, 3
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g)
    s = torch.randn(N, Cin, generator=g) * 0.5 + 1.0
    noise = torch.randn(N, Cout, H, W, generator=g)

    def penalty(f, xx, ww, ss, nz):
        y = f(xx, ww, ss)
        gs, = torch.autograd.grad((y * nz).sum(), ss, create_graph=True)
        return torch.autograd.grad(gs.square().sum(), [ww, ss])

    xr, wr, sr = x.double(), w.double().requires_grad_(True), s.double().requires_grad_(True)
    ref = penalty(lambda a, b, c: ref_modconv.modulated_conv2d(a, b, c, 1, 1, True), xr, wr, sr, noise.double())
    xg, wg, sg = x.cuda(), w.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    got = penalty(lambda a, b, c: gan.functional.modulated_conv2d(a, b, c, None, 1, 1, True, precision=gan._lib.GC_PREC_F32X3),
                  xg, wg, sg, noise.cuda())
    assert rel(got[0], ref[0]) < 5e-4, "second-order gradient by weight"
    assert rel(got[1], ref[1]) < 5e-4, "second-order gradient by style"


def test_modulated_conv_module(gan):
    """The module: parameter names, a bf16-stored activation, and its output against the oracle on its own parameters."""
    torch.manual_seed(0)
    m = gan.nn.ModulatedConv2d(16, 32, 3, style_dim=24).cuda()
    assert sorted(n for n, _ in m.named_parameters()) == ["affine.bias", "affine.weight", "bias", "weight"]
    x = torch.randn(2, 16, 8, 8, device="cuda")
    z = torch.randn(2, 24, device="cuda")
    y = m(x, z)
    style = (z.double().cpu() @ m.affine.weight.double().cpu().t() + m.affine.bias.double().cpu())
    want = ref_modconv.modulated_conv2d(x.double().cpu(), m.weight.double().cpu(), style, 1, 1, True, 1e-8, m.scale) + m.bias.double().cpu().view(1, -1, 1, 1)
    assert rel(y, want) < TOL_BF16
    yb = m(x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last), z)
    assert yb.dtype == torch.bfloat16 and rel(yb, want) < 2e-2
    y.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
