"""Gradients of one critic iteration: the plain one-stream phase against the streamed one, with and without the fused scalar kernels."""
import sys, torch
sys.path.insert(0, '/root/repo')
import gan_amd
from gan_amd.harness import basic_gans as H
for prec, dt in (("fp32x3", torch.float32), ("bf16", torch.bfloat16)):
    gan_amd.set_precision(prec)
    torch.manual_seed(0)
    G = H.Generator(100, 32, 3, 64, activation_dtype=dt).cuda(); D = H.Discriminator(3, 32, 64, is_wgan=True, activation_dtype=dt).cuda()
    G.apply(H.weights_init); D.apply(H.weights_init)
    B = 8
    real = (torch.rand(B, 3, 64, 64, device="cuda") * 2 - 1).contiguous(memory_format=torch.channels_last)
    z = torch.randn(B, 100, 1, 1, device="cuda"); eps = torch.rand(B, 1, 1, 1, device="cuda")
    with torch.no_grad():
        fake = G(z)
    outs = []
    full = set(H.fused_scalars)
    for two, fused in ((False, set()), (True, set()), (True, {"mix"}), (True, {"ones"}), (True, {"penalty"}), (True, {"loss"}), (True, full), (False, full)):
        H.fused_scalars = fused
        gan_amd.ops.clear_cast_cache()
        for p in D.parameters(): p.grad = None
        r = H.wgan_critic_phase(G, D, real, z, eps, 10.0, None, True, two, fake=fake)
        torch.cuda.synchronize()
        outs.append(([p.grad.clone() for p in D.parameters()], [float(x) for x in r]))
    print(prec, [o[1] for o in outs])
    for gs in zip(*[o[0] for o in outs]):
        print("   plain tensor expressions vs streamed with (none, mix, ones, penalty, loss, all) and plain with all:", " ".join(f"{((g - gs[0]).norm() / gs[0].norm()).item():.2e}" for g in gs[1:]))
