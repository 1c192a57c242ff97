"""The weights-resident kernel for 3x3 stride-1 gathers over 64 input channels on large images (gan_amd/csrc/conv_thin.hip):
bit for bit against torch's fp32 convolution on small-integer operands (every product and partial sum exact), forward and
data gradient, both storage types, with bias / LeakyReLU / the gradient mask / per-sample scales in its output stage, on
shapes whose rectangle counts leave workgroups with unequal numbers of rectangles."""
import pytest
import torch

pytestmark = pytest.mark.gpu

THIN = "gc::conv_thin_kernel"


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


def _ints(shape, g, lo, hi, dt=torch.bfloat16):
    return torch.randint(lo, hi, shape, generator=g).float().cuda().to(dt).contiguous(memory_format=torch.channels_last)


CASES = [   # N, Cin, Cout, H, W
    (8, 64, 64, 256, 256),       # 2048 rectangles on 256 sequences
    (10, 64, 64, 256, 224),      # 2240 ... unequal shares, 7 column tiles
    (5, 64, 128, 256, 256),      # two output-channel tiles: 1280 rectangles on 128 sequences
    (3, 64, 256, 256, 256),      # four tiles
    (44, 56, 64, 128, 96),       # 56 input channels (padded to 64 in the weight image), 2112 rectangles
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", CASES, ids=lambda v: str(v))
@pytest.mark.parametrize("out_dt", [torch.float32, torch.bfloat16])
def test_thin_kernel_exact_on_small_integers(gan, N, Cin, Cout, H, W, out_dt):
    F = gan.functional
    if gan._lib.lib().gc_set_patch_kernel(1) == 0:
        gan._lib.lib().gc_set_patch_kernel(1)
    g = torch.Generator().manual_seed(31 + Cout)
    x = _ints((N, Cin, H, W), g, -3, 4).requires_grad_(True)
    w = (torch.randint(-4, 5, (Cout, Cin, 3, 3), generator=g).float() / 8).cuda()
    b = (torch.randint(-8, 9, (Cout,), generator=g).float() / 4).cuda()
    # (an fp32 output with an activation also gets a bf16 copy for the next layer, which this kernel does not write)
    act = "leaky_relu" if out_dt == torch.bfloat16 else None
    y, prof = _profiled(gan, lambda: F.conv2d(x, w, b, 1, 1, activation=act, negative_slope=0.25, out_dtype=out_dt))
    if Cin % 8 == 0:
        assert any(n.startswith(THIN) for n in prof), prof
    yr = torch.nn.functional.conv2d(x.detach().float(), w, b, 1, 1)
    if act is not None:
        yr = torch.nn.functional.leaky_relu(yr, 0.25)
    assert torch.equal(y, yr.to(out_dt))
    if Cout == 64:          # the data gradient of a 64 -> 64 layer reads 64 channels as well
        gy = _ints(tuple(y.shape), g, -2, 3)
        y2 = F.conv2d(x, w, None, 1, 1, out_dtype=torch.bfloat16)
        gx, prof = _profiled(gan, lambda: torch.autograd.grad(y2, x, gy)[0])
        if Cin == 64:
            assert any(n.startswith(THIN) for n in prof), prof
        xr = x.detach().float().requires_grad_(True)
        gxr, = torch.autograd.grad(torch.nn.functional.conv2d(xr, w, None, 1, 1), xr, gy.float())
        assert torch.equal(gx, gxr.to(torch.bfloat16))


def test_thin_kernel_mask_and_scales(gan):
    """The gradient mask (a data gradient that flows on through the LeakyReLU in front of the layer) and the per-sample
    scale tables of the modulated convolution in the kernel's output stage."""
    F, ops = gan.functional, gan.ops
    g = torch.Generator().manual_seed(37)
    N, C, H = 8, 64, 256
    x = _ints((N, C, H, H), g, -3, 4)
    w = (torch.randint(-4, 5, (C, C, 3, 3), generator=g).float() / 8).cuda()
    gy = _ints((N, C, H, H), g, -2, 3)
    act = _ints((N, C, H, H), g, -2, 3)                  # the activated tensor whose sign selects the derivative
    gx, prof = _profiled(gan, lambda: ops.conv_dgrad_masked(gy, w, act, 0.25, 1, 1, gan._lib.GC_PREC_BF16, 0, 1))
    assert any(n.startswith(THIN + "<unsigned short, true>") for n in prof), prof
    xr = x.float().requires_grad_(True)
    gxr, = torch.autograd.grad(torch.nn.functional.conv2d(xr, w, None, 1, 1), xr, gy.float())
    want = gxr * torch.where(act.float() > 0, 1.0, 0.25)
    assert torch.equal(gx, want.to(torch.bfloat16))
    pre = (torch.randint(1, 5, (N, C), generator=g).float() / 2).cuda()
    post = (torch.randint(-4, 5, (N, C), generator=g).float() / 2).cuda()
    bias = (torch.randint(-8, 9, (C,), generator=g).float() / 4).cuda()
    y, prof = _profiled(gan, lambda: ops.modconv_fprop(x, w, bias, pre, post, 1, 1, gan._lib.GC_PREC_BF16, 0, gan._lib.GC_ACT_LRELU, 0.25, 0))
    assert any(n.startswith(THIN) for n in prof), prof
    raw = torch.nn.functional.conv2d(x.float(), w, None, 1, 1)
    want = post.view(N, C, 1, 1) * torch.nn.functional.leaky_relu(raw * pre.view(N, C, 1, 1) + bias.view(1, C, 1, 1), 0.25)
    assert torch.equal(y, want)


def test_small_problems_stay_on_the_patch_kernel(gan):
    """Fewer than eight rectangles per workgroup: the filter load would not pay."""
    g = torch.Generator().manual_seed(41)
    x = _ints((2, 64, 128, 128), g, -3, 4)
    w = (torch.randint(-4, 5, (64, 64, 3, 3), generator=g).float() / 8).cuda()
    y, prof = _profiled(gan, lambda: gan.functional.conv2d(x, w, None, 1, 1))
    assert not any(n.startswith(THIN) for n in prof), prof
    assert torch.equal(y, torch.nn.functional.conv2d(x.float(), w, None, 1, 1).to(y.dtype))
