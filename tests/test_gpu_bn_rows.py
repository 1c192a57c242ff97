"""GPU tests: BatchNorm batch statistics taken from rows of per-channel sums that the convolution in front writes in its
epilogue (gc_epilogue.stat_rows -> gc_batchnorm_act_fwd_rows) against the BatchNorm's own statistics sweep."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gan():
    import gan_amd
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return gan_amd


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _pair(gan, kind, cin, cout, act_dtype):
    gnn = gan.nn
    torch.manual_seed(3)
    if kind == "conv":
        conv = gnn.Conv2d(cin, cout, 4, 2, 1, out_dtype=act_dtype)
    elif kind == "conv_s1":
        conv = gnn.Conv2d(cin, cout, 3, 1, 1, out_dtype=act_dtype)
    elif kind == "convT_first":      # the generators' first layer: a 1 x 1 latent to 4 x 4, odd channel count (register-staged kernel)
        conv = gnn.ConvTranspose2d(cin, cout, 4, 1, 0, bias=False, out_dtype=act_dtype)
    else:
        conv = gnn.ConvTranspose2d(cin, cout, 4, 2, 1, out_dtype=act_dtype)
    return torch.nn.Sequential(conv, gnn.BatchNormAct2d(cout, activation="leaky_relu", negative_slope=0.2)).cuda()


def _run(gan, net, x, fused):
    assert gan.nn.fuse_batchnorm_statistics(net, fused) == 1
    for b in net.buffers():
        b.zero_() if b.dtype != torch.float32 or b.dim() == 0 else b.fill_(0.5)
    net.zero_grad()
    xx = x.clone().requires_grad_(True)
    gan._lib.profile_enable(True)
    y = net(xx)
    torch.cuda.synchronize()
    names = set(gan._lib.profile_collect())
    gan._lib.profile_enable(False)
    (y.float() * torch.linspace(-1, 1, y.numel(), device="cuda").view_as(y)).sum().backward()
    return y, xx.grad, [p.grad.clone() for p in net.parameters()], [b.clone() for b in net.buffers()], names


CASES = [   # producer, Cin, Cout, N, H, storage of the activation
    ("conv", 64, 128, 16, 32, torch.bfloat16),
    ("conv", 64, 64, 8, 64, torch.bfloat16),
    ("conv_s1", 128, 128, 4, 24, torch.bfloat16),
    ("convT", 128, 64, 16, 16, torch.bfloat16),
    ("convT", 256, 128, 32, 8, torch.bfloat16),
    ("conv", 64, 128, 16, 32, torch.float32),
    ("convT_first", 100, 512, 256, 1, torch.bfloat16),
    ("conv", 6, 64, 4, 64, torch.float32),
    # the patch-resident kernel (8 x 32 rectangles: a row of the table per wave), plain, strided and over the four classes
    # of a transposed convolution; large tiles of the LDS-DMA kernel
    ("conv_s1", 128, 128, 32, 64, torch.bfloat16, "gc::conv_patch_kernel<unsigned short, 128, false>"),
    ("conv_s1", 64, 64, 32, 64, torch.bfloat16, "gc::conv_patch_kernel<unsigned short, 64, false>"),
    ("conv", 64, 128, 32, 128, torch.bfloat16, "gc::conv_patch_kernel<unsigned short, 128, false, strided>"),
    ("convT", 128, 64, 32, 32, torch.bfloat16, "gc::conv_patch_kernel<unsigned short, 64, false>"),
    ("convT", 256, 128, 32, 32, torch.bfloat16, "gc::conv_patch_kernel<unsigned short, 128, false>"),
    ("conv_s1", 64, 72, 32, 64, torch.bfloat16, "gc::conv_patch_kernel<unsigned short, 64, false>"),      # a partial channel tile
    ("conv", 256, 512, 64, 32, torch.bfloat16, None),
    ("conv_s1", 512, 512, 32, 32, torch.bfloat16, None),
]


@pytest.mark.parametrize("case", CASES)
def test_batchnorm_from_conv_rows_matches_own_sweep(gan, case):
    kind, cin, cout, N, H, dt = case[:6]
    expect_kernel = case[6] if len(case) > 6 else None
    net = _pair(gan, kind, cin, cout, dt)
    x = torch.randn(N, cin, H, H, device="cuda")
    if cin % 8 == 0:
        x = x.to(dt)
    x = x.contiguous(memory_format=torch.channels_last) * 2 + 0.3
    y0, gx0, gp0, buf0, names0 = _run(gan, net, x, False)
    y1, gx1, gp1, buf1, names1 = _run(gan, net, x, True)
    from_rows = {"gc::bn_rows_partial_kernel", "gc::bn_finalize_kernel<3>"}       # (few rows: the finalizing kernel reads them itself)
    assert "gc::bn_partial_kernel<0>" in names0 and not (from_rows & names0)
    assert (from_rows & names1) and "gc::bn_partial_kernel<0>" not in names1, sorted(names1)
    assert expect_kernel is None or expect_kernel in names1, sorted(names1)
    tol = 2e-5 if dt == torch.float32 else 4e-3         # the statistics agree to rounding; bf16 storage rounds the output
    assert rel(y1, y0) < tol
    assert rel(gx1, gx0) < tol * 5
    for (name, _), a, b in zip(net.named_parameters(), gp1, gp0):
        if name == "0.bias":      # a bias in front of a BatchNorm has no influence: its gradient is rounding noise around zero
            assert bool(torch.isfinite(a).all())
        else:
            assert rel(a, b) < tol * 5, name
    for a, b in zip(buf1, buf0):          # running mean / variance / batch count
        assert rel(a.float(), b.float()) < 1e-5
    y2 = _run(gan, net, x, True)[0]
    assert torch.equal(y1, y2), "rows are reduced in a fixed order"


def test_rows_not_provided_falls_back_to_the_sweep(gan):
    """A producer whose kernel writes no rows (thin fp32 input: the image kernel): the BatchNorm computes its own statistics."""
    gnn = gan.nn
    torch.manual_seed(5)
    net = torch.nn.Sequential(gnn.Conv2d(3, 64, 4, 2, 1), gnn.BatchNormAct2d(64, activation="relu")).cuda()
    x = torch.randn(8, 3, 64, 64, device="cuda")
    y0, _, _, _, names0 = _run(gan, net, x, False)
    y1, _, _, _, names1 = _run(gan, net, x, True)
    assert "gc::bn_partial_kernel<0>" in names1
    assert torch.equal(y0, y1)


def test_statistics_rows_through_the_c_abi(gan):
    """The rows themselves: their fixed-order sum equals the per-channel sums of the stored output."""
    import ctypes
    from gan_amd import ops, _lib
    torch.manual_seed(7)
    x = torch.randn(8, 64, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(128, 64, 4, 4, device="cuda") * 0.05)
    with ops.channel_statistics_request():
        y = ops.conv_fprop(x, w, None, 2, 1, _lib.GC_PREC_BF16, 0)
    ent = ops._row_cache.pop((y.data_ptr(), tuple(y.shape)))
    rows = ent[1][:ent[2] * 128 * 2].view(ent[2], 128, 2).double()
    yf = y.double().permute(0, 2, 3, 1).reshape(-1, 128)
    assert rel(rows[:, :, 0].sum(0), yf.sum(0)) < 1e-6
    assert rel(rows[:, :, 1].sum(0), yf.square().sum(0)) < 1e-6


@pytest.mark.parametrize("kind,cin,cout,N,H", [("up", 128, 64, 32, 32), ("s1", 128, 128, 32, 64), ("up", 256, 128, 16, 16)])
def test_conditional_batchnorm_from_conv_rows_matches_own_sweep(gan, kind, cin, cout, N, H):
    """BigGAN's generator block (BigGAN/model.py:100-133): up-sampling 3x3 convolution -> conditional BatchNorm + ReLU with a
    gain and shift per sample.  With the statistics rows of the convolution the BatchNorm launches no sweep of its own and
    gives the same output, running statistics and saved statistics to rounding."""
    from gan_amd import ops, functional as GF
    torch.manual_seed(17)
    x = (torch.randn(N, cin, H, H, device="cuda") * 2 + 0.3).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)
    b = torch.randn(cout, device="cuda") * 0.1
    gain, shift = torch.randn(N, cout, device="cuda") * 0.3 + 1, torch.randn(N, cout, device="cuda") * 0.2
    res = []
    for rows in (False, True):
        rm, rv, nbt = torch.full((cout,), 0.5, device="cuda"), torch.full((cout,), 0.5, device="cuda"), torch.zeros((), dtype=torch.long, device="cuda")
        gan._lib.profile_enable(True)
        with (ops.channel_statistics_request() if rows else torch.no_grad()):
            h = GF.upsample2x_conv3x3(x, w, b) if kind == "up" else GF.conv2d(x, w, b, 1, 1)
        y = ops.batchnorm_act_train(h, gain, shift, rm, rv, nbt, 0.1, 1e-5, 1, 0.0)
        torch.cuda.synchronize()
        names = set(gan._lib.profile_collect())
        gan._lib.profile_enable(False)
        res.append((y, rm, rv, nbt, names))
    (y0, rm0, rv0, n0, names0), (y1, rm1, rv1, n1, names1) = res
    assert "gc::bn_partial_kernel<0>" in names0
    assert "gc::bn_partial_kernel<0>" not in names1 and ({"gc::bn_rows_partial_kernel", "gc::bn_finalize_kernel<3>"} & names1), sorted(names1)
    assert rel(y1, y0) < 4e-3 and rel(rm1, rm0) < 1e-5 and rel(rv1, rv0) < 1e-5 and int(n1) == int(n0) == 1
