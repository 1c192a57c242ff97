"""Gathers whose input is one value per sample: the data gradient of the critics' score layer, Conv2d(512 -> 1, kernel 4)
on 4 x 4 maps (BasicGANs/models.py:77), and a transposed convolution of a [N, 1, 1, 1] input.  They run as an outer
product (gc::conv_outer_kernel); every output element is one product, so small-integer operands must come out exact,
and random operands equal to the bf16-rounded product."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    return gan_amd


def _names(gan, fn):
    gan._lib.profile_enable(True)
    out = fn()
    torch.cuda.synchronize()
    names = set(gan._lib.profile_collect())
    gan._lib.profile_enable(False)
    return out, names


@pytest.mark.parametrize("cin,hw,N,dt", [(512, 4, 37, torch.bfloat16), (512, 4, 256, torch.float32), (64, 2, 5, torch.bfloat16),
                                         (128, 3, 16, torch.bfloat16)])
@pytest.mark.parametrize("masked", [False, True])
def test_score_layer_data_gradient(gan, cin, hw, N, dt, masked):
    """x -> LeakyReLU epilogue of the layer before (masked: its derivative runs inside the gradient kernel) -> score."""
    g = torch.Generator().manual_seed(61)
    pre = torch.randint(-2, 3, (N, cin, hw, hw), generator=g).float()
    w = torch.randint(-1, 2, (1, cin, hw, hw), generator=g).float()
    gy = torch.randint(-3, 4, (N, 1, 1, 1), generator=g).float()
    slope = 0.25
    pr = pre.double().requires_grad_(True)
    xr = torch.nn.functional.leaky_relu(pr, slope) if masked else pr
    torch.nn.functional.conv2d(xr, w.double()).backward(gy.double())
    want = pr.grad
    xg = pre.cuda().to(dt).contiguous(memory_format=torch.channels_last)
    xg = (torch.nn.functional.leaky_relu(xg, slope) if masked else xg).requires_grad_(True)
    kw = dict(input_activation_slope=slope) if masked else {}
    y = gan.functional.conv2d(xg, w.cuda(), None, out_dtype=torch.float32, **kw)
    (gx,), names = _names(gan, lambda: torch.autograd.grad(y, (xg,), gy.cuda()))
    assert "gc::conv_outer_kernel" in names, sorted(names)
    assert gx.dtype == dt and torch.equal(gx.double().cpu(), want)


def test_score_layer_data_gradient_random_operands(gan):
    """Random operands: each element is the product of the two values rounded to bf16, rounded to the output's type."""
    g = torch.Generator().manual_seed(62)
    x = torch.randn(64, 512, 4, 4, generator=g)
    w = torch.randn(1, 512, 4, 4, generator=g)
    gy = torch.randn(64, 1, 1, 1, generator=g)
    want = gy.bfloat16().double() * w.bfloat16().double()
    xg = x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = gan.functional.conv2d(xg, w.cuda(), None, out_dtype=torch.float32)
    (gx,), names = _names(gan, lambda: torch.autograd.grad(y, (xg,), gy.cuda()))
    assert "gc::conv_outer_kernel" in names, sorted(names)
    assert torch.equal(gx.cpu(), want.float().bfloat16())


@pytest.mark.parametrize("stride,pad,k", [(1, 0, 4), (2, 1, 4), (2, 0, 3)])
def test_transposed_convolution_of_one_value_per_sample(gan, stride, pad, k):
    """ConvTranspose2d(1 -> 64) of a [N, 1, 1, 1] input with bias and activation: pixels no tap reaches hold act(bias)."""
    g = torch.Generator().manual_seed(63)
    N, cout = 21, 64
    x = torch.randint(-3, 4, (N, 1, 1, 1), generator=g).float()
    w = torch.randint(-2, 3, (1, cout, k, k), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    want = torch.nn.functional.leaky_relu(torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), stride, pad), 0.5)
    y, names = _names(gan, lambda: gan.functional.conv_transpose2d(x.cuda(), w.cuda(), b.cuda(), stride, pad, activation="leaky_relu",
                                                                   negative_slope=0.5, out_dtype=torch.float32))
    assert "gc::conv_outer_kernel" in names, sorted(names)
    assert y.shape == want.shape and torch.equal(y.double().cpu(), want)


@pytest.mark.parametrize("cin,cout,hw,N", [(512, 1, 31, 8), (256, 2, 24, 3), (64, 3, 33, 2), (128, 6, 20, 3), (64, 5, 33, 2)])
def test_patch_score_map_on_the_few_channel_kernel(gan, cin, cout, hw, N):
    """The last layer of the PatchGAN discriminator, Conv2d(512 -> 1, kernel 4, stride 1, padding 1) on 31 x 31 maps
    (Pix2Pix/models.py:221-243): the halo of a 4 x 4 filter needs the few-output-channel kernel's 4-row rectangles."""
    g = torch.Generator().manual_seed(64)
    x = torch.randint(-2, 3, (N, cin, hw, hw), generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 4, 4), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1)
    xg = x.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, names = _names(gan, lambda: gan.functional.conv2d(xg, w.cuda(), b.cuda(), 1, 1, out_dtype=torch.float32))
    assert any("conv_scatter_kernel" in n for n in names), sorted(names)
    assert torch.equal(y.double().cpu(), want)


@pytest.mark.parametrize("cin,H,W,k,pad,N,dt", [(512, 31, 31, 4, 1, 5, torch.bfloat16), (512, 31, 31, 4, 1, 3, torch.float32),
                                                (64, 16, 24, 3, 1, 4, torch.bfloat16), (640, 9, 7, 4, 2, 2, torch.bfloat16),
                                                (128, 12, 12, 2, 0, 3, torch.bfloat16)])
def test_sliding_score_layer_data_gradient(gan, cin, H, W, k, pad, N, dt):
    """The data gradient of a score layer that slides over the map (the PatchGAN's Conv2d(512 -> 1, kernel 4, stride 1,
    padding 1), Pix2Pix/models.py:238): one input channel spread over many output channels (gc::conv_spread_kernel).  Small
    integers: every sum is exact, so the result equals torch's CPU gradient bit for bit."""
    g = torch.Generator().manual_seed(63 + cin)
    x = torch.randint(-2, 3, (N, cin, H, W), generator=g).float()
    w = torch.randint(-1, 2, (1, cin, k, k), generator=g).float()
    Ho, Wo = H + 2 * pad - k + 1, W + 2 * pad - k + 1
    gy = torch.randint(-3, 4, (N, 1, Ho, Wo), generator=g).float()
    xr = x.double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w.double(), None, 1, pad).backward(gy.double())
    xg = x.cuda().to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = gan.functional.conv2d(xg, w.cuda(), None, 1, pad, out_dtype=torch.float32)
    (gx,), names = _names(gan, lambda: torch.autograd.grad(y, (xg,), gy.cuda()))
    assert "gc::conv_spread_kernel" in names, sorted(names)
    assert gx.dtype == dt and torch.equal(gx.double().cpu(), xr.grad)
    # the same kernel as the forward pass of a transposed convolution of a one-channel map, with bias and LeakyReLU
    b = torch.randint(-2, 3, (cin,), generator=g).float()
    want = torch.nn.functional.leaky_relu(torch.nn.functional.conv_transpose2d(gy.double(), w.double(), b.double(), 1, pad), 0.25)
    (out,), names = _names(gan, lambda: (gan.functional.conv_transpose2d(gy.cuda().contiguous(memory_format=torch.channels_last), w.cuda(), b.cuda(), 1, pad,
                                                                          activation="leaky_relu", negative_slope=0.25, out_dtype=torch.float32),))
    assert "gc::conv_spread_kernel" in names, sorted(names)
    assert torch.equal(out.double().cpu(), want)


@pytest.mark.parametrize("cimg,cmid,hw,N,dt", [(6, 64, 64, 3, torch.bfloat16), (3, 64, 64, 2, torch.bfloat16), (6, 128, 32, 2, torch.float32)])
def test_image_pair_data_gradient_on_the_few_channel_kernel(gan, cimg, cmid, hw, N, dt):
    """The data gradient of the PatchGAN's first layer, Conv2d(6 -> 64, kernel 4, stride 2, padding 1) on the image pair
    (Pix2Pix/models.py:225-232): 16 taps x 6 channels = 96 (tap, channel) pairs on the few-output-channel kernel.  Small
    integers: bit for bit torch's CPU gradient."""
    g = torch.Generator().manual_seed(65 + cimg)
    x = torch.randint(-2, 3, (N, cimg, hw, hw), generator=g).float()
    w = torch.randint(-1, 2, (cmid, cimg, 4, 4), generator=g).float()
    gy = torch.randint(-2, 3, (N, cmid, hw // 2, hw // 2), generator=g).float()
    xr = x.double().requires_grad_(True)
    torch.nn.functional.conv2d(xr, w.double(), None, 2, 1).backward(gy.double())
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = gan.functional.conv2d(xg, w.cuda(), None, 2, 1, out_dtype=dt)
    (gx,), names = _names(gan, lambda: torch.autograd.grad(y, (xg,), gy.cuda().to(dt)))
    assert any("conv_scatter_kernel" in n for n in names), sorted(names)
    assert torch.equal(gx.double().cpu(), xr.grad)


@pytest.mark.parametrize("cin,cout,H,W,N,act,odt", [(3, 64, 128, 128, 3, "relu", torch.bfloat16), (3, 32, 8, 64, 2, None, torch.float32),
                                                   (4, 48, 12, 32, 5, "leaky_relu", torch.bfloat16), (1, 16, 4, 32, 1, "tanh", torch.float32)])
def test_image_layer_3x3(gan, cin, cout, H, W, N, act, odt):
    """A 3 x 3 / stride 1 / padding 1 convolution on an fp32 image of at most 4 channels (BigGAN's first discriminator block,
    BigGAN/model.py:181-200) on the image-layer kernel.  Small integers: every sum is exact, so the pre-activation values
    equal torch's CPU result bit for bit."""
    g = torch.Generator().manual_seed(71 + cout)
    x = torch.randint(-3, 4, (N, cin, H, W), generator=g).float()
    w = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 1, 1)
    if act == "relu":
        want = torch.relu(want)
    elif act == "leaky_relu":
        want = torch.nn.functional.leaky_relu(want, 0.25)
    elif act == "tanh":
        want = torch.tanh(want)
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    y, names = _names(gan, lambda: gan.functional.conv2d(xg, w.cuda(), b.cuda(), 1, 1, activation=act, negative_slope=0.25, out_dtype=odt))
    assert "gc::conv_image_kernel" in names, sorted(names)
    assert y.dtype == odt
    if act == "tanh":
        assert torch.allclose(y.double().cpu(), want, atol=2e-6)
    else:
        assert torch.equal(y.double().cpu(), want)          # (multiples of 1/4 of small integers: exact in bf16 too)


@pytest.mark.parametrize("k", [3, 1])
@pytest.mark.parametrize("cin,cout,H,W,N,xdt", [(3, 64, 128, 128, 5, torch.float32), (3, 32, 8, 64, 3, torch.float32), (4, 48, 12, 32, 7, torch.bfloat16),
                                                (1, 16, 4, 32, 2, torch.float32)])
def test_image_layer_weight_gradient_stride_1(gan, cin, cout, H, W, N, xdt, k):
    """Weight and bias gradient of a 3 x 3 / padding 1 (BigGAN/model.py:181-200) or 1 x 1 (a from-RGB layer,
    StyleGAN/model.py:453-514) stride-1 layer on an image, on the image-layer weight-gradient kernel; small integers, bit for
    bit torch's CPU gradient."""
    g = torch.Generator().manual_seed(75 + cout + k)
    x = torch.randint(-3, 4, (N, cin, H, W), generator=g).float()
    w = torch.randint(-2, 3, (cout, cin, k, k), generator=g).float()
    b = torch.zeros(cout)
    gy = torch.randint(-2, 3, (N, cout, H, W), generator=g).float()
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    torch.nn.functional.conv2d(x.double(), wr, br, 1, k // 2).backward(gy.double())
    xg = x.cuda().to(xdt).contiguous(memory_format=torch.channels_last)
    wg, bg = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = gan.functional.conv2d(xg, wg, bg, 1, k // 2, out_dtype=torch.bfloat16)
    (gw, gb), names = _names(gan, lambda: torch.autograd.grad(y, (wg, bg), gy.cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)))
    assert "gc::conv_wgrad_image_kernel" in names and not any("copy" in n for n in names), sorted(names)
    assert torch.equal(gw.double().cpu(), wr.grad) and torch.equal(gb.double().cpu(), br.grad)
