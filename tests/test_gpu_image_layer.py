"""Forward pass of the critics' first layer, Conv2d(<= 4 channels -> 16 .. 128 or 5 .. 8 channels -> 16 .. 64 (the PatchGAN's image pair, Pix2Pix/models.py:221-243), kernel 4, stride 2, padding 1) on fp32 images
(BasicGANs/models.py:52-56,77), on its own kernel (gc::conv_image_kernel): every output-channel count the kernel takes (its
channel blocks are interleaved in pairs, a last block without a partner is not), both output types, fused and plain
activations.  Small-integer operands: every product and sum is exact, so the result must equal torch's on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    return gan_amd


@pytest.mark.parametrize("cin,cout", [(3, 64), (3, 48), (1, 32), (4, 16), (2, 64), (3, 128), (3, 96), (4, 112), (6, 64), (5, 32), (8, 48)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("activation", [None, "leaky_relu", "tanh"])
def test_image_layer_forward_is_exact_on_small_integers(gan, cin, cout, out_dtype, activation):
    g = torch.Generator().manual_seed(71 + cin + cout)
    N = 5
    x = torch.randint(-2, 3, (N, cin, 64, 64), generator=g).float()
    w = torch.randint(-1, 2, (cout, cin, 4, 4), generator=g).float()
    b = torch.randint(-3, 4, (cout,), generator=g).float()
    want = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), 2, 1)
    if activation == "leaky_relu":
        want = torch.nn.functional.leaky_relu(want, 0.25)
    elif activation == "tanh":
        want = torch.tanh(want)
    kw = {} if activation is None else dict(activation=activation, negative_slope=0.25)
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    gan._lib.profile_enable(True)
    y = gan.functional.conv2d(xg, w.cuda(), b.cuda(), 2, 1, out_dtype=out_dtype, **kw)
    torch.cuda.synchronize()
    names = set(gan._lib.profile_collect())
    gan._lib.profile_enable(False)
    if out_dtype == torch.bfloat16 or activation is None:      # (an activated fp32 output may come with a bf16 copy: another kernel)
        assert "gc::conv_image_kernel" in names, sorted(names)
    assert y.dtype == out_dtype and y.shape == want.shape
    if activation == "tanh":
        tol = 2e-6 if out_dtype == torch.float32 else 4e-3
        assert (y.double().cpu() - want).abs().max().item() <= tol
    elif out_dtype == torch.float32:
        assert torch.equal(y.double().cpu(), want)
    else:       # sums of up to 64 small products and quarter multiples: exact in bf16 when below 256 in magnitude
        assert torch.equal(y.cpu(), want.float().bfloat16())
