"""The patch kernel's strided mode: the forward pass of a 4x4 / stride 2 / padding 1 convolution (the down-sampling layers of
the StyleGAN critic, StyleGAN/model.py:246-262, and of the Pix2Pix encoder, Pix2Pix/model.py:23-44) run as the 2x2-tap
stride-1 convolution over the 2x2-block image, with nothing rearranged in memory.  Small-integer operands make every product
and partial sum exact in bf16 x bf16 -> fp32, so the results must equal torch's fp32 convolution bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

STRIDED = "gc::conv_patch_kernel"


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
    (16, 128, 256, 128, 128),    # two 128-channel tiles
    (32, 96, 192, 128, 128),     # 3 chunks of 32; the second channel tile is half empty
    (32, 80, 128, 128, 128),     # 80 channels: the third chunk of every sub-pixel is half padding
    (8, 64, 64, 256, 256),       # the 64-channel tile
    (9, 64, 320, 128, 192),      # three tiles, unequal extents
    (16, 512, 128, 128, 256),    # 16 chunks per sub-pixel
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", CASES, ids=lambda v: str(v))
@pytest.mark.parametrize("out_dt", [torch.float32, torch.bfloat16])
def test_strided_mode_exact_on_small_integers(gan, N, Cin, Cout, H, W, out_dt):
    F = gan.functional
    gan._lib.lib().gc_set_patch_kernel(1)
    g = torch.Generator().manual_seed(41 + Cin)
    x = _ints((N, Cin, H, W), g, -3, 4)
    w = (torch.randint(-4, 5, (Cout, Cin, 4, 4), generator=g).float() / 8).cuda()
    b = (torch.randint(-8, 9, (Cout,), generator=g).float() / 4).cuda()
    act = "leaky_relu" if out_dt == torch.bfloat16 else None
    y, prof = _profiled(gan, lambda: F.conv2d(x, w, b, 2, 1, activation=act, negative_slope=0.25, out_dtype=out_dt))
    assert any(n.startswith(STRIDED) and "strided" in n for n in prof), prof
    yr = torch.nn.functional.conv2d(x.float(), w, b, 2, 1)
    if act is not None:
        yr = torch.nn.functional.leaky_relu(yr, 0.25)
    assert torch.equal(y, yr.to(out_dt))


def test_strided_mode_border_pixels_and_channel_slice(gan):
    """An image of ones under a filter of ones counts the taps that fall inside the image: 16 in the interior, 12 on an edge,
    9 in a corner, times the channel count - any sub-pixel fetched from the wrong side of a border shows up as a wrong count."""
    F = gan.functional
    N, C, H, W = 16, 64, 128, 128
    x = torch.ones(N, C, H, W, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.ones(128, C, 4, 4, device="cuda")
    y, prof = _profiled(gan, lambda: F.conv2d(x, w, None, 2, 1, out_dtype=torch.float32))
    yr = torch.nn.functional.conv2d(x.float(), w, None, 2, 1)
    assert torch.equal(y, yr)
    assert y[0, 0, 0, 0].item() == 9 * C and y[0, 0, 0, 5].item() == 12 * C and y[0, 0, 7, 7].item() == 16 * C
    assert y[0, 0, -1, -1].item() == 9 * C and y[0, 0, -1, 5].item() == 12 * C


def test_strided_mode_matches_tiled_kernels_on_random_data(gan):
    """Random bf16 data: the strided mode against the tiled kernels (same products, another summation order)."""
    F = gan.functional
    g = torch.Generator().manual_seed(43)
    x = torch.randn(16, 128, 128, 128, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 128, 4, 4, generator=g) * 0.03).cuda()
    b = torch.randn(256, generator=g).cuda()
    y, prof = _profiled(gan, lambda: F.conv2d(x, w, b, 2, 1, activation="leaky_relu", negative_slope=0.2, out_dtype=torch.bfloat16))
    assert any("strided" in n for n in prof), prof
    prev = gan._lib.lib().gc_set_patch_kernel(0)
    try:
        y2, prof2 = _profiled(gan, lambda: F.conv2d(x, w, b, 2, 1, activation="leaky_relu", negative_slope=0.2, out_dtype=torch.bfloat16))
    finally:
        gan._lib.lib().gc_set_patch_kernel(prev)
    assert not any("strided" in n for n in prof2)
    # fp32 sums in two orders, each rounded once to bf16: equal up to one unit in the last place of a few elements
    d = (y.float() - y2.float()).abs()
    assert d.max().item() <= 2 ** -7 * y2.float().abs().max().item() and (d > 0).float().mean().item() < 0.02
