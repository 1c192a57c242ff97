"""Developer probe: time of BigGAN's orthogonal regularisation over both networks (ops.orthogonal_regularization_)."""
import sys
sys.path.insert(0, ".")
import torch
import gan_amd
from gan_amd import ops
from gan_amd.harness import biggan

torch.manual_seed(0)
G, D = biggan.Generator(n_classes=10, ch=64).cuda(), biggan.Discriminator(n_classes=10, ch=64).cuda()
for net in (G, D):
    for p in net.parameters():
        p.grad = torch.zeros_like(p)
ops.set_kernel_timing(True) if hasattr(ops, "set_kernel_timing") else None
for name, net in (("G", G), ("D", D)):
    for _ in range(2):
        biggan.orthogonal_regularization(net)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        biggan.orthogonal_regularization(net)
    b.record()
    torch.cuda.synchronize()
    n = sum(1 for p in net.parameters() if p.dim() > 2)
    print(name, "%d weights, %.3f ms per call" % (n, a.elapsed_time(b) / 5), flush=True)
    want = [p.grad.clone() for p in net.parameters() if p.dim() > 2]
    for p in net.parameters():
        p.grad.zero_()
    for p in net.parameters():
        if p.dim() > 2:
            w = p.detach().flatten(1).double()
            g = w @ w.t()
            g.fill_diagonal_(0.0)
            p.grad.add_((2e-4 * (g @ w)).view_as(p).float())
    ref = [p.grad.clone() for p in net.parameters() if p.dim() > 2]
    worst = max(((x / 7 - y).norm() / y.norm().clamp_min(1e-30)).item() for x, y in zip(want, ref))
    print(name, "largest relative error of the added term over the weights: %.2e" % worst, flush=True)
    for p in net.parameters():
        p.grad.zero_()


@torch.no_grad()
def with_library_gemms(params, beta=1e-4):
    for p in params:
        wb = p.detach().flatten(1).to(torch.bfloat16)
        gram = wb @ wb.t()
        gram.fill_diagonal_(0.0)
        p.grad.add_((gram @ wb).view_as(p), alpha=2.0 * beta)


for name, net in (("G", G), ("D", D)):
    params = [p for p in net.parameters() if p.dim() > 2]
    for _ in range(2):
        with_library_gemms(params)
    torch.cuda.synchronize()
    for p in params:
        p.grad.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        with_library_gemms(params)
    b.record()
    torch.cuda.synchronize()
    print(name, "library GEMMs on bf16 copies: %.3f ms per call" % (a.elapsed_time(b) / 5), flush=True)
    worst = 0.0
    for p in params:
        w = p.detach().flatten(1).double()
        g = w @ w.t()
        g.fill_diagonal_(0.0)
        y = (2e-4 * (g @ w)).view_as(p)
        worst = max(worst, ((p.grad.double() / 5 - y).norm() / y.norm().clamp_min(1e-30)).item())
    print(name, "largest relative error: %.2e" % worst, flush=True)
    # as one HIP graph (what the recorded training step pays)
    for fn, label in ((lambda: biggan.orthogonal_regularization(net), "convolution kernels"), (lambda: with_library_gemms(params), "library GEMMs")):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        ops.clear_weight_cache(); ops.clear_cast_cache()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g), ops.capturing_on():
            fn()
        g.replay(); torch.cuda.synchronize()
        a.record()
        for _ in range(10):
            g.replay()
        b.record()
        torch.cuda.synchronize()
        print(name, label, "recorded: %.3f ms per call" % (a.elapsed_time(b) / 10), flush=True)
