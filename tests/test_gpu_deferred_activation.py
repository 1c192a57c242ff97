"""GPU tests of the LeakyReLU derivative moved out of sweeps of its own: into the blur's backward sweep (gc_dwblur3x3_masked)
and into the data-gradient kernel of the next convolution, as the StyleGAN critic blocks use them
(StyleGAN/model.py:409-450: conv + LeakyReLU -> blur -> strided conv + LeakyReLU -> next block).  The checker is the same
arithmetic with the activation's own backward sweep (torch autograd over the library's plain operators), so results agree
to rounding of the stored gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gan_amd


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [(2, 8, 5, 7), (3, 64, 32, 32), (1, 12, 40, 3), (2, 16, 1, 1)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("slope", [0.2, 0.0])
def test_masked_blur_against_torch(gan, shape, dt, slope):
    g = torch.Generator().manual_seed(7)
    N, C, H, W = shape
    gy = torch.randn(shape, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    act = torch.randn(shape, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    got = gan.ops.blur3x3_masked(gy, act, slope)
    k = torch.tensor([1.0, 2.0, 1.0], dtype=torch.float64)
    k = (k[:, None] * k[None, :] / 16).view(1, 1, 3, 3).repeat(C, 1, 1, 1)
    want = torch.nn.functional.conv2d(gy.double().cpu(), k, None, 1, 1, 1, C) * torch.where(act.double().cpu() > 0, 1.0, slope)
    assert got.dtype == dt
    assert rel(got, want) < (1e-6 if dt == torch.float32 else 4e-3)
    plain = gan.ops.blur3x3(gy)
    if dt == torch.float32:       # the same sums, then one product
        assert torch.equal(got, plain * torch.where(act > 0, 1.0, slope).to(dt))
    # the adjoint: the derivative applied to what the blur reads
    pre = gan.ops.blur3x3_masked(gy, act, slope, True)
    want_pre = torch.nn.functional.conv2d(gy.double().cpu() * torch.where(act.double().cpu() > 0, 1.0, slope), k, None, 1, 1, 1, C)
    assert pre.dtype == dt and rel(pre, want_pre) < (1e-6 if dt == torch.float32 else 4e-3)
    u = torch.randn(shape, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    lhs, rhs = (got.double() * u.double()).sum(), (gy.double() * gan.ops.blur3x3_masked(u, act, slope, True).double()).sum()
    assert abs(float(lhs - rhs)) < (1e-5 if dt == torch.float32 else 2e-2) * (abs(float(lhs)) + 1.0)


def _pair(gan, x, w1, b1, w2, b2, fused, code):
    """conv 3x3 + LeakyReLU -> blur -> conv 4x4 stride 2 + LeakyReLU -> conv 3x3 (the next block's first layer)."""
    F = gan.functional
    h = F.conv2d(x, w1, b1, 1, 1, precision=code, activation="leaky_relu", negative_slope=0.2, defer_activation_grad=fused)
    h = F.blur3x3(h, 0.2 if fused else None)
    h = F.conv2d(h, w2, b2, 2, 1, precision=code, activation="leaky_relu", negative_slope=0.2, defer_activation_grad=fused)
    return F.conv2d(h, w1, None, 1, 1, precision=code, input_activation_slope=0.2 if fused else None)


@pytest.mark.parametrize("dt,code_name,tol", [(torch.float32, "GC_PREC_F32X3", 2e-5), (torch.bfloat16, "GC_PREC_BF16", 1.5e-2)])
def test_deferred_chain_first_and_second_order(gan, dt, code_name, tol):
    """Gradients by input, weights and biases, and a gradient penalty's double backward, with and without the deferral."""
    code = getattr(gan._lib, code_name)
    g = torch.Generator().manual_seed(9)
    N, C, H = 4, 64, 32
    x = torch.randn(N, C, H, H, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    w1 = (torch.randn(C, C, 3, 3, generator=g) / 24).cuda()
    w2 = (torch.randn(C, C, 4, 4, generator=g) / 32).cuda()
    b1, b2 = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.randn(C, generator=g) * 0.1).cuda()
    seed = torch.randn(N, C, H // 2, H // 2, generator=g).to(dt).cuda()
    res = []
    for fused in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        y = _pair(gan, *leaves, fused, code)
        first = torch.autograd.grad((y * seed).sum(), leaves, create_graph=True)
        penalty = (first[0].float().flatten(1).norm(dim=1) - 1).square().mean()
        second = torch.autograd.grad(penalty, leaves[1:], allow_unused=True)
        res.append((y, first, second))
    (ya, fa, sa), (yb, fb, sb) = res
    assert torch.equal(ya, yb)                       # the forward pass is untouched
    for a, b, name in zip(fb, fa, ("input", "w1", "b1", "w2", "b2")):
        assert rel(a, b) < tol, f"gradient by {name}: {rel(a, b)}"
    for a, b, name in zip(sb, sa, ("w1", "b1", "w2", "b2")):
        if b is None or float(b.float().norm()) == 0.0:
            assert a is None or float(a.float().abs().max()) < 1e-6, name
        else:
            assert rel(a, b) < 4 * tol, f"second-order gradient by {name}: {rel(a, b)}"


def test_deferred_chain_launches(gan):
    """With the deferral the backward pass of the pair launches no activation sweep of its own."""
    code = gan._lib.GC_PREC_BF16
    g = torch.Generator().manual_seed(11)
    N, C, H = 4, 64, 32
    x = torch.randn(N, C, H, H, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    w1, w2 = (torch.randn(C, C, 3, 3, generator=g) / 24).cuda(), (torch.randn(C, C, 4, 4, generator=g) / 32).cuda()
    b1, b2 = torch.zeros(C).cuda(), torch.zeros(C).cuda()
    leaves = [t.requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = _pair(gan, *leaves, True, code)
    gan._lib.profile_enable(True)
    try:
        torch.autograd.grad(y.float().sum(), leaves)
        torch.cuda.synchronize()
        prof = gan._lib.profile_collect()
    finally:
        gan._lib.profile_enable(False)
    assert "gc::dwblur3x3_masked_kernel" in prof and "gc::dwblur3x3_kernel" not in prof, prof
    assert "gc::act_backward_sum_kernel" not in prof, prof


def test_stylegan_critic_with_and_without_the_deferral(gan):
    """The critic at 32 x 32 (progressive step 3): scores, parameter gradients and the gradient-penalty step agree between the
    fused chain (deferred derivatives) and the plain one."""
    from gan_amd.harness import stylegan as S
    torch.manual_seed(3)
    D = S.Discriminator().cuda()
    x = torch.randn(4, 3, 32, 32, device="cuda")
    out = []
    try:
        for fused in (False, True):
            S.set_deferred_activations(fused)
            D.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            score = D(xx, 3, 1)
            (slope,) = torch.autograd.grad(score.sum(), xx, create_graph=True)
            pen = (slope.flatten(1).norm(dim=1) - 1).square().mean()
            (score.mean() + 10 * pen).backward()
            out.append((score.detach(), {n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None}))
    finally:
        S.set_deferred_activations(True)
    (sa, ga), (sb, gb) = out
    assert rel(sb, sa) < 1e-5          # (the small layers' split reductions add in an order that varies from run to run)
    assert ga.keys() == gb.keys()
    worst = max((rel(gb[n], ga[n]), n) for n in ga if float(ga[n].norm()) > 0)
    assert worst[0] < 2e-2, worst


def test_biggan_discriminator_blocks_with_and_without_the_deferral(gan):
    """BigGAN's relu-conv-relu-conv blocks (BigGAN/model.py:180-203): scores and every parameter gradient agree between the
    form whose ReLU derivatives ride on the data-gradient kernels and the form with their own sweeps."""
    from gan_amd.harness import biggan as B
    torch.manual_seed(5)
    D = B.Discriminator(n_classes=4, ch=8).cuda()
    x = torch.randn(4, 3, 128, 128, device="cuda")
    y = torch.randint(0, 4, (4,), device="cuda")
    out = []
    try:
        for fused in (False, True):
            B.set_deferred_activations(fused)
            D.zero_grad(set_to_none=True)
            for m in D.modules():                       # the same power-iteration state in both runs
                if hasattr(m, "U_vector"):
                    torch.manual_seed(11)
                    m.U_vector.copy_(torch.rand_like(m.U_vector))
            score = D(x, y)
            torch.relu(1 - score).mean().backward()
            out.append((score.detach(), {n: p.grad.clone() for n, p in D.named_parameters() if p.grad is not None}))
    finally:
        B.set_deferred_activations(True)
    (sa, ga), (sb, gb) = out
    assert rel(sb, sa) < 1e-5
    assert ga.keys() == gb.keys() and len(ga) > 20
    worst = max((rel(gb[n], ga[n]), n) for n in ga if float(ga[n].norm()) > 0)
    assert worst[0] < 2e-2, worst


@pytest.mark.parametrize("cin,dt", [(3, torch.bfloat16), (3, torch.float32), (64, torch.bfloat16)])
def test_second_order_mask_leaves_with_the_producing_kernel(gan, cin, dt):
    """conv + LeakyReLU (derivative deferred) -> conv, under a gradient penalty's double backward (the critic's first two layers,
    BasicGANs/models.py:77-79): the second-order pass needs (derivative of the activation) x (forward convolution of the incoming
    vector).  With ops.fuse_second_order_mask the forward convolution's kernel applies that factor in its output stage and the
    consumer skips its sweep: same gradients to rounding, one elementwise launch less."""
    from torch.profiler import ProfilerActivity, profile
    F = gan.functional
    g = torch.Generator().manual_seed(13)
    N, H, C1, C2 = 8, 64, 64, 128
    x = (torch.randn(N, cin, H, H, generator=g)).cuda().contiguous(memory_format=torch.channels_last)
    if cin % 8 == 0:
        x = x.to(dt)
    w1 = (torch.randn(C1, cin, 4, 4, generator=g) / (4 * cin ** 0.5)).cuda()
    w2 = (torch.randn(C2, C1, 4, 4, generator=g) / 32).cuda()
    seed = torch.randn(N, C2, H // 4, H // 4, generator=g).to(dt).cuda()
    res = []
    was = gan.ops.fuse_second_order_mask
    try:
        for fuse in (False, True):
            gan.ops.fuse_second_order_mask = fuse
            xx, a, b = (t.clone().requires_grad_(True) for t in (x, w1, w2))
            h = F.conv2d(xx, a, None, 2, 1, activation="leaky_relu", negative_slope=0.2, out_dtype=dt, defer_activation_grad=True)
            y = F.conv2d(h, b, None, 2, 1, out_dtype=dt, input_activation_slope=0.2)
            gx, = torch.autograd.grad((y * seed).sum(), xx, create_graph=True)
            penalty = (gx.float().flatten(1).norm(dim=1) - 1).square().mean()
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                second = torch.autograd.grad(penalty, (a, b))
                torch.cuda.synchronize()
            sweeps = sum("leaky_relu_backward" in e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
            res.append((gx.detach(), second, sweeps))
    finally:
        gan.ops.fuse_second_order_mask = was
    (gxa, sa, na), (gxb, sb, nb) = res
    assert torch.equal(gxa, gxb)                      # the first-order pass is untouched
    assert na >= 1 and nb == na - 1, (na, nb)
    tol = 2e-5 if (dt == torch.float32 and cin == 3) else 1.5e-2
    for p, q, name in zip(sb, sa, ("w1", "w2")):
        assert rel(p, q) < tol, f"second-order gradient by {name}: {rel(p, q)}"
