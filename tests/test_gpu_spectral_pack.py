"""GPU tests: packed weight images divided by a device scalar at pack time (gc_pack_weights_div, ops.repack_divided) -- the
spectral norm's weight / sigma (BigGAN/model.py:20-35) without the quotient tensor -- against the convolution called with
the quotient tensor itself: the same bits, forward and backward."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gan_amd


@pytest.mark.parametrize("precision", ["bf16", "fp32x3"])
def test_divided_images_equal_the_images_of_the_quotient(gan, precision):
    ops, F = gan.ops, gan.functional
    gan.set_precision(precision)
    try:
        torch.manual_seed(11)
        layer = gan.nn.Conv2d(8, 16, 3, padding=1, bias=False).cuda()
        key = layer._wkey_for(layer.weight)
        x = torch.randn(3, 8, 12, 12, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        seed = torch.randn(3, 16, 12, 12, device="cuda").contiguous(memory_format=torch.channels_last)
        for sig in (torch.tensor([3.0, 1.7], device="cuda"), torch.tensor([0.5, 0.9], device="cuda")):     # first use, then a refresh
            y_ref = F.conv2d(x, layer.weight / sig[1], None, 1, 1)
            (gx_ref,) = torch.autograd.grad(y_ref, x, seed)
            epoch = ops.repack_divided([key], [layer.weight], sig, [1])
            assert ops.divisor_epoch(key) == epoch
            y = F.conv2d(x, layer.weight, None, 1, 1, _wkey=key)
            (gx,) = torch.autograd.grad(y, x, seed)
            assert torch.equal(y, y_ref) and torch.equal(gx, gx_ref)
        with torch.no_grad():
            layer.weight.mul_(1.25)                     # the weights move: the next registration packs the new values
        sig = torch.tensor([2.0, 1.1], device="cuda")
        ops.repack_divided([key], [layer.weight], sig, [1], forward_only=True)
        y = F.conv2d(x, layer.weight, None, 1, 1, _wkey=key)
        (gx,) = torch.autograd.grad(y, x, seed)          # (the data-gradient image was left out above: refreshed at this use)
        y_ref = F.conv2d(x, layer.weight / sig[1], None, 1, 1)
        (gx_ref,) = torch.autograd.grad(y_ref, x, seed)
        assert torch.equal(y, y_ref) and torch.equal(gx, gx_ref)
    finally:
        gan.set_precision("bf16")


def test_biggan_critic_with_and_without_divided_images(gan):
    """One forward / backward pass of the reduced-width discriminator in both modes from the same state: the same bits.  A
    backward pass through a forward pass that is no longer the layers' latest is refused."""
    from gan_amd.harness import biggan
    torch.manual_seed(12)
    D1 = biggan.Discriminator(n_classes=4, ch=8).cuda()
    D2 = copy.deepcopy(D1)
    x = torch.randn(4, 3, 128, 128, device="cuda").contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 4, (4,), device="cuda")
    try:
        biggan.set_divided_pack(True)
        out1 = D1(x, y)
        out1.sum().backward()
        biggan.set_divided_pack(False)
        out2 = D2(x, y)
        out2.sum().backward()
        assert torch.equal(out1, out2)
        for (n, p), q in zip(D1.named_parameters(), D2.parameters()):
            assert (p.grad is None) == (q.grad is None), n
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), n
        biggan.set_divided_pack(True)
        first = D1(x, y)
        D1(x, y)
        with pytest.raises(RuntimeError, match="another forward pass"):
            first.sum().backward()
    finally:
        biggan.set_divided_pack(True)


def test_recorded_biggan_step_matches_eager_steps(gan):
    """A reduced-width BigGAN training step recorded as one HIP graph (the recording keeps the packed images of the warm-up,
    every forward pass refreshes them with its sigmas in one launch) against the same steps run eagerly: the loss terms of
    the third and fourth step."""
    from gan_amd.harness import steps as S
    try:
        eager = S.biggan_128("cuda:0", batch=4, ch=8, n_classes=4)
        want = []
        for _ in range(4):                       # (the eager twin runs first and alone)
            eager()
            want.append(eager.losses.clone())
        torch.cuda.synchronize()
        del eager
        gan.ops.clear_weight_cache()
        step = S.biggan_128("cuda:0", batch=4, ch=8, n_classes=4)
        assert step.weight_images_refreshed_in_step
        g = gan.GraphedStep(step, warmup=2)      # steps one and two
        got = []
        for _ in range(2):
            g()
            got.append(step.losses.clone())
        torch.cuda.synchronize()
        for a, b in zip(got, want[2:]):
            assert bool(torch.isfinite(a).all())
            assert torch.allclose(a, b, rtol=2e-3, atol=1e-5), (a, b)
    finally:
        S.reset_options()
