"""1 x 1 convolutions of images with at most four channels (the from-RGB layers of the StyleGAN discriminator,
StyleGAN/model.py:453-514) on their own kernel (gc::conv_point_kernel): exact on small integers against torch on the CPU,
within bf16 rounding on random operands, gradients included (the kernel leaves the backward passes to the others)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    return gan_amd


def _run(gan, fn):
    gan._lib.profile_enable(True)
    out = fn()
    torch.cuda.synchronize()
    names = set(gan._lib.profile_collect())
    gan._lib.profile_enable(False)
    return out, names


@pytest.mark.parametrize("cin,cout,N,H,W", [(3, 64, 2, 256, 256), (3, 32, 5, 128, 128), (1, 8, 4, 128, 130), (4, 128, 1, 256, 258), (2, 256, 3, 160, 144)])
@pytest.mark.parametrize("in_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_point_convolution_is_exact_on_small_integers(gan, cin, cout, N, H, W, in_dtype, out_dtype):
    g = torch.Generator().manual_seed(81 + cin)
    x = torch.randint(-3, 4, (N, cin, H, W), generator=g).float()
    w = torch.randint(-2, 3, (cout, cin, 1, 1), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    want = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, w, b), 0.25)
    xg = x.cuda().to(in_dtype).contiguous(memory_format=torch.channels_last)
    y, names = _run(gan, lambda: gan.functional.conv2d(xg, w.cuda(), b.cuda(), activation="leaky_relu", negative_slope=0.25, out_dtype=out_dtype))
    if out_dtype == torch.bfloat16:
        assert "gc::conv_point_kernel" in names, sorted(names)
    assert y.dtype == out_dtype and torch.equal(y.float().cpu(), want)


def test_point_convolution_with_scale_and_gradients(gan):
    """Random operands with the equalized-learning-rate constant folded into the packed weight, as the harness calls it;
    input, weight and bias gradients against float64 on the CPU."""
    g = torch.Generator().manual_seed(82)
    x = torch.randn(4, 3, 128, 128, generator=g)
    w = torch.randn(64, 3, 1, 1, generator=g)
    b = torch.randn(64, generator=g)
    gy = torch.randn(4, 64, 128, 128, generator=g)
    c = (2.0 / 3) ** 0.5
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.conv2d(xr, wr * c, br)
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, wr, br), gy.double())
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y, names = _run(gan, lambda: gan.functional.conv2d(xg, wg, bg, weight_scale=c, out_dtype=torch.bfloat16))
    assert "gc::conv_point_kernel" in names, sorted(names)
    gx, gw, gb = torch.autograd.grad(y, (xg, wg, bg), gy.cuda().to(torch.bfloat16))

    def rel(a, r):
        return ((a.double().cpu() - r).norm() / r.norm()).item()
    assert rel(y, yr) < 6e-3 and rel(gx, gxr) < 8e-3 and rel(gw, gwr) < 8e-3 and rel(gb, gbr) < 8e-3
