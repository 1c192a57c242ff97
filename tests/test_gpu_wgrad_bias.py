"""The bias gradient summed inside the weight-gradient launch (a fragment of ones next to the x fragments in the LDS-DMA and
row-streaming kernels; per-split partial sums added in split order by the reduction launch): exact on small integers, equal
from run to run, no pass of its own over gy, and differentiable once more."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    gan_amd.set_precision("bf16")
    return gan_amd


def _profiled(gan, fn):
    gan._lib.profile_enable(True)
    try:
        out = fn()
        torch.cuda.synchronize()
        prof = gan._lib.profile_collect()
    finally:
        gan._lib.profile_enable(False)
    return out, prof


def _ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)


CASES = [   # N, Cin, Cout, H, W, k, stride, pad, kernel expected in the profile
    (3, 64, 64, 32, 32, 3, 1, 1, "gc::conv_wgrad_rows_kernel"),
    (5, 128, 192, 8, 64, 3, 1, 1, "gc::conv_wgrad_rows_kernel"),         # three output-channel tiles, two input-channel tiles
    (40, 64, 64, 32, 32, 3, 1, 1, "gc::conv_wgrad_rows_kernel"),         # several strips per split
    (4, 64, 128, 16, 16, 4, 2, 1, "gc::conv_wgrad_dma_kernel"),          # strided 4x4 (critic down-sampling)
    (6, 128, 128, 16, 16, 3, 1, 1, "gc::conv_wgrad_dma_kernel"),         # image narrower than a strip
    (2, 512, 512, 8, 8, 3, 1, 1, "gc::conv_wgrad_dma_kernel"),           # wide layer, few pixels
    (3, 64, 72, 12, 12, 3, 1, 1, "gc::conv_wgrad_dma_kernel"),           # output channels not a multiple of 16
    (2, 40, 24, 10, 10, 1, 1, 0, "gc::conv_wgrad_dma_kernel"),           # 1x1
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(str(v) for v in c[:8]))
def test_bias_gradient_from_the_weight_gradient_launch(gan, case):
    N, Cin, Cout, H, W, k, stride, pad, kernel = case
    ops = gan.ops
    g = torch.Generator().manual_seed(5 + Cout)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x, gy = _ints((N, Cin, H, W), g), _ints((N, Cout, Ho, Wo), g)
    (gw, gb), prof = _profiled(gan, lambda: ops.conv_wgrad_bias(x, gy, k, k, stride, pad, gan._lib.GC_PREC_BF16))
    assert any(n.startswith(kernel) for n in prof), prof
    assert not any("colsum" in n or "channel_sum" in n for n in prof), prof
    want_b = gy.float().sum((0, 2, 3))
    assert torch.equal(gb, want_b)                                           # small integers: every partial sum is exact
    w = torch.zeros(Cout, Cin, k, k, device="cuda", requires_grad=True)
    want_w, = torch.autograd.grad(torch.nn.functional.conv2d(x.float(), w, None, stride, pad), w, gy.float())
    assert torch.equal(gw, want_w)
    assert torch.equal(gw, ops.conv_wgrad(x, gy, k, k, stride, pad, gan._lib.GC_PREC_BF16))
    # random data: the same bits from run to run
    xr = torch.randn(N, Cin, H, W, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    gr = torch.randn(N, Cout, Ho, Wo, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
    a = ops.conv_wgrad_bias(xr, gr, k, k, stride, pad, gan._lib.GC_PREC_BF16)
    b = ops.conv_wgrad_bias(xr, gr, k, k, stride, pad, gan._lib.GC_PREC_BF16)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    ref = gr.double().sum((0, 2, 3))
    assert float((a[1].double() - ref).norm() / ref.norm()) < 1e-5


def test_bias_gradient_falls_back_to_the_channel_sum(gan):
    """fp32 operands (three-term split) and thin inputs are not summed in the kernel: same values from channel_sum."""
    ops = gan.ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 8, 8, generator=g).cuda()
    gy = torch.randn(2, 8, 8, 8, generator=g).cuda()
    (gw, gb), prof = _profiled(gan, lambda: ops.conv_wgrad_bias(x, gy, 3, 3, 1, 1, gan._lib.GC_PREC_F32X3))
    assert any("channel_sum" in n for n in prof), prof
    assert torch.allclose(gb, gy.sum((0, 2, 3)), rtol=1e-5, atol=1e-5)


def test_conv_backward_with_bias_uses_one_launch_and_differentiates_twice(gan):
    """functional.conv2d with a bias on bf16 activations: the backward pass holds no channel sum, gradients match torch, and
    a gradient penalty through it (double backward of the bias-summing weight gradient) matches torch's."""
    F = gan.functional
    g = torch.Generator().manual_seed(13)
    N, Cin, Cout, H = 4, 64, 64, 32
    x = torch.randn(N, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 24
    b = torch.randn(Cout, generator=g) * 0.1
    seed = torch.randn(N, Cout, H, H, generator=g)
    xg = x.to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = F.conv2d(xg, wg, bg, 1, 1, precision=gan._lib.GC_PREC_BF16)
    (gx, gw, gb), prof = _profiled(gan, lambda: torch.autograd.grad((y * seed.cuda().to(y.dtype)).sum(), [xg, wg, bg], create_graph=True))
    assert not any("colsum" in n or "channel_sum" in n for n in prof), prof
    pen = gw.square().sum() + gb.square().sum()
    ggx, = torch.autograd.grad(pen, xg, allow_unused=True)      # d pen / d x through the weight gradient only
    xr = x.to(torch.bfloat16).double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    sr = seed.to(torch.bfloat16).double()
    yr = torch.nn.functional.conv2d(xr, wr, br, 1, 1)
    rx, rw, rb = torch.autograd.grad((yr * sr).sum(), [xr, wr, br], create_graph=True)
    rggx, = torch.autograd.grad(rw.square().sum() + rb.square().sum(), xr, allow_unused=True)

    def rel(a, b_):
        a, b_ = a.detach().double().cpu(), b_.detach().double().cpu()
        return float((a - b_).norm() / b_.norm().clamp_min(1e-30))
    assert rel(gb, rb) < 1e-5 and rel(gw, rw) < 1e-2 and rel(gx, rx) < 1e-2
    assert rel(ggx, rggx) < 2e-2
