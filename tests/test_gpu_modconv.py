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


# ---- the fused form: demodulation factor and the next layer's style in the convolution's output stage -----------------
def _profiled(gan, fn):
    gan._lib.profile_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        prof = gan._lib.profile_collect()
    finally:
        gan._lib.profile_enable(False)
    return out, prof


def _chain_reference(x, ws, bs, ss, pads, slope, c):
    """Three modulated convolutions with bias and LeakyReLU, each the textbook per-sample-weight form."""
    h = x
    for w, b, s, pad, ci in zip(ws, bs, ss, pads, c):
        h = ref_modconv.modulated_conv2d(h, w, s, 1, pad, True, 1e-8, ci) + b.view(1, -1, 1, 1)
        h = torch.nn.functional.leaky_relu(h, slope)
    return h


def _chain_fused(gan, x, ws, bs, ss, pads, slope, c, code, out_dtype=None):
    F = gan.functional
    h = gan.ops.channel_scale(x, ss[0])            # the first layer's input has no modulated producer
    for i, (w, b, s, pad, ci) in enumerate(zip(ws, bs, ss, pads, c)):
        h = F.modulated_conv2d(h, w, s, b, 1, pad, True, 1e-8, weight_scale=ci, precision=code, activation="leaky_relu",
                               negative_slope=slope, input_modulated=True, next_style=ss[i + 1] if i + 1 < len(ws) else None,
                               out_dtype=out_dtype)
    return h


CHAINS = [   # N, channels of the four tensors, H, W, kernel sizes
    (3, (8, 12, 16, 8), 9, 7, (3, 1, 3)),           # ragged extents: the generic tiled kernel
    (4, (64, 128, 64, 64), 16, 16, (3, 3, 1)),      # full channel tiles on small images: the LDS-DMA tiled kernel
    (16, (64, 64, 64, 64), 128, 128, (3, 3, 3)),    # 8 x 32 rectangles, >= 512 tiles: the patch-resident kernel
]


@pytest.mark.parametrize("chain", CHAINS)
@pytest.mark.parametrize("prec,tol", [("fp32x3", 2e-4), ("bf16", 2e-2)])
def test_fused_chain_against_oracle_parity_unpinned(gan, chain, prec, tol):
    """A chain of three modulated convolutions, every layer ONE launch (the convolution kernel applies the demodulation
    factor, bias, LeakyReLU and the next layer's style): output and the gradients by input, weights, biases and all three
    styles against the per-sample-weight formulation in float64.  Parity unpinned: the reference has no such layer."""
    N, ch, H, W, ks = chain
    if prec == "fp32x3" and N * H * W > 100000:
        pytest.skip("the float64 checker of the large case runs once, in the bf16 mode that reaches the patch kernel")
    g = torch.Generator().manual_seed(17 + ch[0])
    x = torch.randn(N, ch[0], H, W, generator=g)
    ws = [torch.randn(ch[i + 1], ch[i], k, k, generator=g) for i, k in enumerate(ks)]
    bs = [torch.randn(ch[i + 1], generator=g) * 0.3 for i in range(3)]
    ss = [torch.randn(N, ch[i], generator=g) * 0.4 + 1.0 for i in range(3)]
    pads = [k // 2 for k in ks]
    c = [1.0 / (ch[i] * k * k) ** 0.5 for i, k in enumerate(ks)]
    slope = 0.2
    dev_ref = "cuda" if N * H * W > 100000 else "cpu"          # float64 on the GPU's stock torch operators for the large case
    leaves_r = [t.double().to(dev_ref).requires_grad_(True) for t in [x] + ws + bs + ss]
    yr = _chain_reference(leaves_r[0], leaves_r[1:4], leaves_r[4:7], leaves_r[7:10], pads, slope, c)
    seed = torch.randn(yr.shape, generator=g)
    gr = torch.autograd.grad(yr, leaves_r, seed.double().to(dev_ref))
    code = {"fp32x3": gan._lib.GC_PREC_F32X3, "bf16": gan._lib.GC_PREC_BF16}[prec]
    leaves_g = [t.cuda().requires_grad_(True) for t in [x] + ws + bs + ss]
    if prec == "bf16":
        leaves_g[0] = x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    yg, prof = _profiled(gan, lambda: _chain_fused(gan, leaves_g[0], leaves_g[1:4], leaves_g[4:7], leaves_g[7:10], pads, slope, c, code))
    names = prof
    convs = [n for n in names if n.startswith("gc::conv_")]
    assert sum(prof[n]["launches"] for n in convs) == 3, names
    assert prof["gc::channel_scale_kernel"]["launches"] == 1, names             # the sweep in front of the first layer only
    if prec == "bf16" and chain[0] == 16:
        assert all(n.startswith("gc::conv_patch_kernel") for n in convs), names
    if prec == "bf16" and chain[0] == 4:
        assert all(n.startswith("gc::conv_gather_dma_kernel") for n in convs), names
    gg, prof = _profiled(gan, lambda: torch.autograd.grad(yg, leaves_g, seed.cuda().to(yg.dtype)))
    assert prof["gc::mod_act_backward_kernel"]["launches"] == 3, prof
    assert prof["gc::channel_scale_kernel"]["launches"] == 1, prof              # the adjoint of the one forward sweep
    assert rel(yg, yr) < tol, "output"
    what = ["input"] + [f"weight {i}" for i in range(3)] + [f"bias {i}" for i in range(3)] + [f"style {i}" for i in range(3)]
    for a, b, name in zip(gg, gr, what):
        # bf16 mode: activations and gradients are STORED in bf16 here, three layers deep, with LeakyReLU kinks in between
        bound = (2 * tol if name.startswith("bias") else tol) if prec == "fp32x3" else 4 * tol
        assert rel(a, b) < bound, f"gradient by {name}: {rel(a, b)}"


def test_fused_chain_second_order_parity_unpinned(gan):
    """Path-length-style penalty through two fused layers: gradient of |d y / d style_0|^2 by weights and styles (the backward
    sweep differentiated once more)."""
    N, ch, H, W = 2, (8, 8, 8), 6, 6
    g = torch.Generator().manual_seed(23)
    x = torch.randn(N, ch[0], H, W, generator=g)
    ws = [torch.randn(ch[i + 1], ch[i], 3, 3, generator=g) for i in range(2)]
    bs = [torch.randn(ch[i + 1], generator=g) * 0.3 for i in range(2)]
    ss = [torch.randn(N, ch[i], generator=g) * 0.4 + 1.0 for i in range(2)]
    noise = torch.randn(N, ch[2], H, W, generator=g)
    c = [1.0 / (ch[i] * 9) ** 0.5 for i in range(2)]

    def penalty(f, leaves, nz):
        y = f(leaves)
        gs, = torch.autograd.grad((y * nz).sum(), leaves[5], create_graph=True)
        return torch.autograd.grad(gs.square().sum(), leaves[1:])

    lr = [t.double().requires_grad_(True) for t in [x] + ws + bs + ss]
    ref = penalty(lambda l: _chain_reference(l[0], l[1:3], l[3:5], l[5:7], (1, 1), 0.2, c), lr, noise.double())
    lg = [t.cuda().requires_grad_(True) for t in [x] + ws + bs + ss]
    got = penalty(lambda l: _chain_fused(gan, l[0], l[1:3], l[3:5], l[5:7], (1, 1), 0.2, c, gan._lib.GC_PREC_F32X3), lg, noise.cuda())
    for a, b, name in zip(got, ref, ["weight 0", "weight 1", "bias 0", "bias 1", "style 0", "style 1"]):
        if float(b.norm()) == 0.0:      # the penalty does not depend on the biases (piecewise-linear layers)
            assert float(a.abs().max()) < 1e-5, f"second-order gradient by {name}"
        else:
            assert rel(a, b) < 1e-3, f"second-order gradient by {name}: {rel(a, b)}"


@pytest.mark.parametrize("geom", [(16, 64, 64, 128, 128, "gc::conv_patch_kernel"), (16, 128, 256, 64, 64, "gc::conv_patch_kernel"),
                                  (4, 64, 128, 16, 16, "gc::conv_gather_dma_kernel")])
def test_fused_layer_equals_the_three_launch_composition(gan, geom):
    """bf16 operands, fp32 storage: the fused layer against the same arithmetic composed of the library's plain operators
    (channel_scale, conv2d, channel_scale, bias, LeakyReLU, channel_scale) -- identical operand rounding, so output and all
    gradients agree to fp32 rounding, on the geometries of the patch-resident and the LDS-DMA tiled kernels."""
    N, Cin, Cout, H, W, kernel = geom
    F, ops = gan.functional, gan.ops
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, Cin, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).cuda()
    b = (torch.randn(Cout, generator=g) * 0.3).cuda()
    s = (torch.randn(N, Cin, generator=g) * 0.4 + 1.0).cuda()
    s2 = (torch.randn(N, Cout, generator=g) * 0.4 + 1.0).cuda()
    seed = torch.randn(N, Cout, H, W, generator=g).cuda()
    c = 1.0 / (Cin * 9) ** 0.5
    code = gan._lib.GC_PREC_BF16

    def fused(x, w, b, s, s2):
        return F.modulated_conv2d(x, w, s, b, 1, 1, True, 1e-8, weight_scale=c, precision=code, activation="leaky_relu",
                                  next_style=s2, out_dtype=torch.float32)

    def composed(x, w, b, s, s2):
        y = F.conv2d(ops.channel_scale(x, s), w, None, 1, 1, precision=code, out_dtype=torch.float32, weight_scale=c)
        d = torch.rsqrt(s.square() @ (w * c).square().sum((2, 3)).t() + 1e-8)
        y = torch.nn.functional.leaky_relu(ops.channel_scale(y, d) + b.view(1, -1, 1, 1), 0.2)
        return ops.channel_scale(y, s2)

    la = [t.clone().requires_grad_(True) for t in (x, w, b, s, s2)]
    lb = [t.clone().requires_grad_(True) for t in (x, w, b, s, s2)]
    ya, prof = _profiled(gan, lambda: fused(*la))
    assert any(n.startswith(kernel) for n in prof), prof
    yb = composed(*lb)
    ga = torch.autograd.grad(ya, la, seed)
    gb = torch.autograd.grad(yb, lb, seed)
    assert rel(ya, yb) < 1e-5
    for a_, b_, name in zip(ga, gb, ("input", "weight", "bias", "style", "next_style")):
        assert rel(a_, b_) < 2e-3, f"gradient by {name}: {rel(a_, b_)}"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("slope", [1.0, 0.0, 0.2])
def test_mod_act_backward_against_torch(gan, dt, slope):
    """The backward sweep alone: gc, dot and sum against the tensor expressions; several slabs per sample (64 x 64)."""
    g = torch.Generator().manual_seed(31)
    N, C, H, W = 3, 24, 64, 64
    gy = torch.randn(N, C, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    out = torch.randn(N, C, H, W, generator=g).to(dt).cuda().contiguous(memory_format=torch.channels_last)
    pre = (torch.rand(N, C, generator=g) + 0.5).cuda()
    post = torch.randn(N, C, generator=g).cuda()
    gc, dot, sm = gan.ops.mod_act_backward(gy, out, pre, post, slope)
    gd, od = gy.double(), out.double()
    po, pr = post.double().view(N, C, 1, 1), pre.double().view(N, C, 1, 1)
    m = torch.where(po * od > 0, 1.0, slope)
    gu = gd * po * m
    assert gc.dtype == dt
    assert rel(gc, gu * pr) < (1e-6 if dt == torch.float32 else 4e-3)
    assert rel(dot, (gd * od).sum((2, 3))) < 2e-6
    assert rel(sm, gu.sum((2, 3))) < 2e-6
    again = gan.ops.mod_act_backward(gy, out, pre, post, slope)
    assert all(torch.equal(a, b) for a, b in zip((gc, dot, sm), again))          # fixed summation order
    gc1, dot1, sm1 = gan.ops.mod_act_backward(gy, out, None, None, slope)
    m1 = torch.where(od > 0, 1.0, slope)
    assert rel(gc1, gd * m1) < (1e-6 if dt == torch.float32 else 4e-3) and rel(sm1, (gd * m1).sum((2, 3))) < 2e-6


def test_sample_scales_through_the_c_abi_on_a_data_gradient(gan):
    """gc_epilogue.sample_scale / sample_scale_out on gc_conv2d_dgrad (a transposed convolution's forward pass), which no
    Python caller uses: gx = post[n, c] * lrelu(pre[n, c] * dgrad + bias[c]) against torch in float64."""
    import ctypes
    L = gan._lib
    N, Cin, Cout, H, W, k, stride, pad = 3, 16, 24, 8, 8, 4, 2, 1
    Ho = (H + 2 * pad - k) // stride + 1
    g = torch.Generator().manual_seed(37)
    gy = torch.randn(N, Cout, Ho, Ho, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, generator=g).cuda()
    bias = torch.randn(Cin, generator=g).cuda()
    pre, post = (torch.rand(N, Cin, generator=g) + 0.5).cuda(), torch.randn(N, Cin, generator=g).cuda()
    d = gan.ops._desc(N, H, W, Cin, Ho, Ho, Cout, k, k, stride, pad, L.GC_F32, L.GC_PREC_F32X3, L.GC_F32)
    wp = gan.ops._packed(d, w, L.GC_PASS_DGRAD, 0, 1.0)
    gx = torch.empty(N, Cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    ep = L.Epilogue()
    ep.act, ep.slope, ep.bias = L.GC_ACT_LRELU, 0.2, bias.data_ptr()
    ep.sample_scale, ep.sample_scale_out = pre.data_ptr(), post.data_ptr()
    L.check(L.lib().gc_conv2d_dgrad(ctypes.byref(d), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                    ctypes.c_void_p(gx.data_ptr()), ctypes.byref(ep), None), "gc_conv2d_dgrad")
    torch.cuda.synchronize()
    raw = torch.nn.functional.conv_transpose2d(gy.double(), w.double(), None, stride, pad)
    want = post.double().view(N, Cin, 1, 1) * torch.nn.functional.leaky_relu(raw * pre.double().view(N, Cin, 1, 1) + bias.double().view(1, -1, 1, 1), 0.2)
    assert rel(gx, want) < TOL_X3
    # the combinations the output stage does not offer are refused
    ep.ch_sum = bias.data_ptr()
    assert L.lib().gc_conv2d_dgrad(ctypes.byref(d), ctypes.c_void_p(gy.data_ptr()), ctypes.c_void_p(wp.data_ptr()),
                                   ctypes.c_void_p(gx.data_ptr()), ctypes.byref(ep), None) == -5         # GC_BAD_ARGUMENT
