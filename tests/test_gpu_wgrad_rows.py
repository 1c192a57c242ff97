"""The row-streaming weight-gradient kernel for 3x3 / stride 1 / padding 1 layers (gan_amd/csrc/conv_wgrad_rows.hip)
through the C ABI: bit for bit against torch's fp32 weight gradient on small-integer operands (every product and every
partial sum is exact in bf16 x bf16 -> fp32, so the result does not depend on tiling, split or summation order), on
shapes that exercise every border of its indexing: one and several column strips, strips at both image borders, images
shorter than the ring, several samples per split and several splits per sample, channel tiles on both sides, channel
slices of wider buffers, several (x, gy) segments in one launch; and against the tiled kernel it replaces on random
data."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS = "gc::conv_wgrad_rows_kernel"


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


def _ref_wgrad(x, gy, k=3, p=1):
    w = torch.zeros(gy.shape[1], x.shape[1], k, k, dtype=torch.float32, device=x.device, requires_grad=True)
    y = torch.nn.functional.conv2d(x.float(), w, None, stride=1, padding=p)
    gw, = torch.autograd.grad(y, w, gy.float())
    return gw


CASES = [
    # N, Cin, Cout, H, W
    (3, 64, 64, 32, 32),      # one strip per image
    (5, 128, 64, 8, 64),      # two strips, both touch a border; two input-channel tiles
    (2, 64, 128, 64, 32),     # two output-channel tiles, tall strips
    (2, 192, 128, 16, 96),    # three strips (an interior one), 3 x 2 tiles
    (1, 64, 64, 4, 32),       # fewer rows than ring slots
    (7, 64, 64, 5, 128),      # odd height, four strips
    (40, 64, 64, 32, 32),     # more strips than splits ... several strips per workgroup
    (1, 256, 256, 32, 64),    # 16 tiles
]


@pytest.mark.parametrize("N,Cin,Cout,H,W", CASES, ids=lambda v: str(v))
def test_rows_kernel_exact_on_small_integers(gan, N, Cin, Cout, H, W):
    from gan_amd import ops
    g = torch.Generator().manual_seed(41)
    x = torch.randint(-3, 4, (N, Cin, H, W), generator=g).float().cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randint(-2, 3, (N, Cout, H, W), generator=g).float().cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gw, prof = _profiled(gan, lambda: ops.conv_wgrad(x, gy, 3, 3, 1, 1, gan._lib.GC_PREC_BF16))
    assert ROWS in prof, sorted(prof)
    assert torch.equal(gw, _ref_wgrad(x, gy))


def test_rows_kernel_reads_channel_slices(gan):
    """x and gy as channel slices of wider NHWC buffers (what a U-Net's concatenation buffers hand the kernels)."""
    from gan_amd import ops
    g = torch.Generator().manual_seed(42)
    N, H, W = 3, 16, 64
    xw = torch.randint(-3, 4, (N, H, W, 64 + 128 + 8), generator=g).float().cuda().to(torch.bfloat16)
    gyw = torch.randint(-2, 3, (N, H, W, 16 + 64), generator=g).float().cuda().to(torch.bfloat16)
    x = xw[..., 64:192].permute(0, 3, 1, 2)       # logical NCHW views with NHWC strides and a channel offset
    gy = gyw[..., 16:80].permute(0, 3, 1, 2)
    gw, prof = _profiled(gan, lambda: ops.conv_wgrad(x, gy, 3, 3, 1, 1, gan._lib.GC_PREC_BF16))
    if ROWS not in prof:       # the Python layer may have made dense copies; the result must be right either way
        pass
    assert torch.equal(gw, _ref_wgrad(x, gy))


@pytest.mark.parametrize("Ns", [(2, 1), (3, 3, 1, 2), (1, 6)])
def test_rows_kernel_over_segments(gan, Ns):
    """gc_conv2d_wgrad_multi: several (x, gy) pairs of a layer in one launch, strips dealt out across the segments."""
    from gan_amd import ops
    Cin, Cout, H, W = 64, 128, 8, 64
    torch.manual_seed(43)
    pairs = []
    for n in Ns:
        x = torch.randint(-3, 4, (n, Cin, H, W), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = torch.randint(-3, 4, (n, Cout, H, W), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        pairs.append((x, gy))
    want = sum(_ref_wgrad(x, gy) for x, gy in pairs)
    got = torch.full_like(want, float("nan"))
    _, prof = _profiled(gan, lambda: ops._wgrad_launch_multi(pairs, got, 3, 3, 1, 1, gan._lib.GC_PREC_BF16, 1.0, accumulate=False))
    assert ROWS in prof, sorted(prof)
    assert torch.equal(got, want)
    ops._wgrad_launch_multi(pairs, got, 3, 3, 1, 1, gan._lib.GC_PREC_BF16, 0.5, accumulate=True)
    assert torch.equal(got, 1.5 * want)


def test_rows_kernel_against_float64_on_random_data(gan):
    """Random bf16 operands at a StyleGAN layer's shape (128 -> 256 at 64 x 64, batch 4): the fp32 sums of products of
    bf16 values against the same sums in float64; 2e-5 is the tolerance of the other bf16-exact kernel tests."""
    from gan_amd import ops
    torch.manual_seed(44)
    x = torch.randn(4, 128, 64, 64, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(4, 256, 64, 64, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gw, prof = _profiled(gan, lambda: ops.conv_wgrad(x, gy, 3, 3, 1, 1, gan._lib.GC_PREC_BF16))
    assert ROWS in prof
    w = torch.zeros(256, 128, 3, 3, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x.cpu().double(), w, None, stride=1, padding=1)
    ref, = torch.autograd.grad(y, w, gy.cpu().double())
    err = (gw.cpu().double() - ref).norm() / ref.norm()
    assert err < 2e-5, err


def test_rows_kernel_declines_what_it_cannot_run(gan):
    """Widths that are no multiple of 32, channel counts that are no multiple of 64 and other filters stay on the tiled
    kernels (and stay exact)."""
    from gan_amd import ops
    g = torch.Generator().manual_seed(45)
    for (N, Cin, Cout, H, W) in [(2, 64, 64, 16, 48), (2, 96, 64, 16, 32), (2, 64, 72, 16, 32)]:
        x = torch.randint(-3, 4, (N, Cin, H, W), generator=g).float().cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gy = torch.randint(-2, 3, (N, Cout, H, W), generator=g).float().cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        gw, prof = _profiled(gan, lambda: ops.conv_wgrad(x, gy, 3, 3, 1, 1, gan._lib.GC_PREC_BF16))
        assert ROWS not in prof
        assert torch.equal(gw, _ref_wgrad(x, gy))
