"""Nearest x2 replication folded into the 3 x 3 convolution behind it (gan_amd.functional.upsample2x_conv3x3): the
generator blocks of the reference's BigGAN (BigGAN/model.py:116-120) and the low-resolution synthesis blocks of its
StyleGAN (StyleGAN/model.py:236-238), against torch's interpolate + conv2d on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    return gan_amd


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def _reference(x, w, b, act):
    y = torch.nn.functional.conv2d(torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    if act == "relu":
        y = torch.relu(y)
    elif act == "leaky_relu":
        y = torch.nn.functional.leaky_relu(y, 0.25)
    return y


@pytest.mark.parametrize("N,Cin,H,W,Cout,act", [(3, 64, 8, 8, 128, None), (2, 128, 16, 12, 64, "relu"), (5, 72, 5, 7, 24, "leaky_relu"),
                                                 (2, 16, 4, 4, 8, None), (1, 256, 32, 32, 256, None)])
def test_folded_replication_is_exact_on_small_integers(gan, N, Cin, H, W, Cout, act):
    """Small-integer operands: every folded tap (a sum of up to four taps), product and partial sum is exact, so output,
    input gradient, weight gradient and bias gradient must equal torch's bit for bit."""
    g = torch.Generator().manual_seed(41)
    x = torch.randint(-2, 3, (N, Cin, H, W), generator=g).float()
    w = torch.randint(-1, 2, (Cout, Cin, 3, 3), generator=g).float()
    b = torch.randint(-3, 4, (Cout,), generator=g).float()
    gy = torch.randint(-2, 3, (N, Cout, 2 * H, 2 * W), generator=g).float()
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = _reference(xr, wr, br, act)
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), gy)
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    kw = {} if act is None else dict(activation=act, negative_slope=0.25)
    y = gan.functional.upsample2x_conv3x3(xg, wg, bg, out_dtype=torch.float32, **kw)
    gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), gy.cuda())
    assert y.shape == yr.shape
    assert torch.equal(y.cpu(), yr) and torch.equal(gx.cpu(), gxr) and torch.equal(gw.cpu(), gwr) and torch.equal(gb.cpu(), gbr)


@pytest.mark.parametrize("prec,tol", [("fp32x3", 2e-5), ("bf16", 8e-3)])
def test_folded_replication_against_torch(gan, prec, tol):
    """Random operands, both precisions, with the equalized-learning-rate constant folded in as StyleGAN's layers use it."""
    gan.set_precision(prec)
    try:
        g = torch.Generator().manual_seed(42)
        x = torch.randn(4, 96, 16, 16, generator=g)
        w = torch.randn(128, 96, 3, 3, generator=g)
        b = torch.randn(128, generator=g)
        c = (2.0 / (96 * 9)) ** 0.5
        gy = torch.randn(4, 128, 32, 32, generator=g)
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        yr = _reference(xr, wr * c, br, None)
        gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), gy)
        xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        y = gan.functional.upsample2x_conv3x3(xg, wg, bg, weight_scale=c, out_dtype=torch.float32)
        gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), gy.cuda())
        assert rel(y, yr) < tol and rel(gx, gxr) < tol and rel(gw, gwr) < tol and rel(gb, gbr) < tol
    finally:
        gan.set_precision("bf16")


def test_one_by_one_shortcut_commutes_with_replication(gan):
    """BigGAN/model.py:131-132 runs the block's 1 x 1 shortcut on the up-scaled input; the harness runs it on the small input
    and replicates its output: the same numbers."""
    g = torch.Generator().manual_seed(43)
    x = torch.randn(2, 64, 8, 8, generator=g).cuda()
    w = torch.randn(32, 64, 1, 1, generator=g).cuda()
    b = torch.randn(32, generator=g).cuda()
    a = gan.functional.conv2d(gan.functional.upsample_nearest2x(x), w, b)
    c = gan.functional.upsample_nearest2x(gan.functional.conv2d(x, w, b))
    assert torch.equal(a, c)


# ---- the weight-space fold kernel itself (gc_fused_upsample_weight) ---------------------------------------------------
_M = {0: torch.tensor([[1.0, 0, 0], [1, 1, 0], [0, 1, 1], [0, 0, 1]]) / 2, 1: torch.tensor([[0.0, 0, 1], [0, 1, 1], [1, 1, 0], [1, 0, 0]])}


@pytest.mark.parametrize("fold", [0, 1])
@pytest.mark.parametrize("swap", [False, True])
def test_weight_fold_and_its_adjoint_against_torch(fold, swap):
    """fold 0 against the reference's own expression (pad to 5 x 5, mean of the four one-step shifts:
    StyleGAN/model.py:396-402), fold 1 against the 4 x 3 matrix form; the adjoint through <fold(w), g> = <w, unfold(g)> and
    through autograd, twice."""
    import gan_amd
    ops = gan_amd.ops
    g = torch.Generator().manual_seed(3 + fold)
    A, B = 37, 20
    w = torch.randn((B, A, 3, 3) if swap else (A, B, 3, 3), generator=g).cuda().requires_grad_(True)
    w4 = ops.upsample_weight_fold(w, fold, swap)
    assert w4.shape == (A, B, 4, 4)
    wt = w.detach().transpose(0, 1) if swap else w.detach()
    m = _M[fold].cuda()
    want = torch.einsum("ar,xyrs,bs->xyab", m, wt, m)
    assert torch.allclose(w4, want, rtol=1e-6, atol=1e-6)
    if fold == 0:
        p = torch.nn.functional.pad(wt, [1, 1, 1, 1])
        ref = (p[:, :, 1:, 1:] + p[:, :, :-1, 1:] + p[:, :, 1:, :-1] + p[:, :, :-1, :-1]) / 4
        assert torch.allclose(w4, ref, rtol=1e-6, atol=1e-6)
    g4 = torch.randn(A, B, 4, 4, generator=g).cuda().requires_grad_(True)
    g3 = ops.upsample_weight_unfold(g4, fold, swap)
    assert g3.shape == w.shape
    lhs, rhs = (w4 * g4).sum(), (w * g3).sum()
    assert abs(float((lhs - rhs).detach())) < 1e-3 * abs(float(lhs.detach())) + 1e-3
    gw, = torch.autograd.grad((w4 * g4).sum(), w, create_graph=True)           # = unfold(g4), still a function of g4
    assert torch.allclose(gw, g3.detach(), rtol=1e-6, atol=1e-6)
    gg, = torch.autograd.grad((gw * w.detach()).sum(), g4)                     # second order: the fold again
    assert torch.allclose(gg, w4.detach(), rtol=1e-6, atol=1e-6)


def test_weight_fold_multiplier_through_the_c_abi():
    import ctypes
    import gan_amd
    L = gan_amd._lib
    w = torch.randn(8, 5, 3, 3, device="cuda")
    out = torch.empty(8, 5, 4, 4, device="cuda")
    L.check(L.lib().gc_fused_upsample_weight(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()), 8, 5, 45, 9, 0.25, 0, 0, None), "fold")
    torch.cuda.synchronize()
    m = _M[0].cuda()
    assert torch.allclose(out, 0.25 * torch.einsum("ar,xyrs,bs->xyab", m, w, m), rtol=1e-6, atol=1e-6)
    assert L.lib().gc_fused_upsample_weight(ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(out.data_ptr()), 8, 5, 45, 9, 1.0, 0, 2, None) == -5
