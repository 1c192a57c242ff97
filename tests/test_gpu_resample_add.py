"""GPU tests: 2x2 pooling / 2x replication with a residual block's sum in the same sweep (gc_pool2x2_add,
gc_replicate2x_add; BigGAN/model.py:131-133, 196-200) against the unfused pair of torch operators on the CPU."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gan_amd


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 16, 8, 12), (2, 64, 32, 32), (1, 8, 2, 2)])
def test_fused_sums_are_exact_on_small_integers(gan, dt, shape):
    F = gan.functional
    g = torch.Generator().manual_seed(5)
    N, C, H, W = shape
    big = torch.randint(-4, 5, (N, C, 2 * H, 2 * W), generator=g).float()
    small = torch.randint(-4, 5, (N, C, H, W), generator=g).float()
    bg = big.to(dt).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sg = small.to(dt).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    seed_small = torch.randint(-2, 3, (N, C, H, W), generator=g).float()
    seed_big = torch.randint(-2, 3, (N, C, 2 * H, 2 * W), generator=g).float()
    # pooled big + small
    y = F.avg_pool2x2_add(bg, sg)
    want = torch.nn.functional.avg_pool2d(big.double(), 2) + small.double()
    assert y.dtype == dt and torch.equal(y.double().cpu(), want)          # (quarters of small integers are exact in bf16 here)
    gb, gs = torch.autograd.grad(y, (bg, sg), seed_small.to(dt).cuda())
    assert torch.equal(gs.double().cpu(), seed_small.double())
    assert torch.equal(gb.double().cpu(), torch.nn.functional.interpolate(seed_small.double(), scale_factor=2, mode="nearest") * 0.25)
    # up-sampled small + big
    z = F.upsample_nearest2x_add(sg, bg)
    want = torch.nn.functional.interpolate(small.double(), scale_factor=2, mode="nearest") + big.double()
    assert z.dtype == dt and torch.equal(z.double().cpu(), want)
    gs, gb = torch.autograd.grad(z, (sg, bg), seed_big.to(dt).cuda())
    assert torch.equal(gb.double().cpu(), seed_big.double())
    assert torch.equal(gs.double().cpu(), torch.nn.functional.avg_pool2d(seed_big.double(), 2) * 4)


def test_fallback_for_channel_counts_the_kernel_does_not_take(gan):
    F = gan.functional
    x = torch.randn(2, 3, 8, 8, device="cuda")
    r = torch.randn(2, 3, 4, 4, device="cuda")
    assert torch.allclose(F.avg_pool2x2_add(x, r), torch.nn.functional.avg_pool2d(x, 2) + r)
    assert torch.allclose(F.upsample_nearest2x_add(r, x), torch.nn.functional.interpolate(r, scale_factor=2) + x)
