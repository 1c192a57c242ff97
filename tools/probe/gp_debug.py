"""Where the fused gradient-penalty kernel and the tensor expression part ways in bf16 mode (debug aid)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gan_amd
from gan_amd import ops
from gan_amd.harness import basic_gans as H
gan_amd.set_precision("bf16")
torch.manual_seed(0)
dt = torch.bfloat16
G = H.Generator(100, 32, 3, 64, activation_dtype=dt).cuda(); D = H.Discriminator(3, 32, 64, is_wgan=True, activation_dtype=dt).cuda()
G.apply(H.weights_init); D.apply(H.weights_init)
B = 8
real = (torch.rand(B, 3, 64, 64, device="cuda") * 2 - 1).contiguous(memory_format=torch.channels_last)
z = torch.randn(B, 100, 1, 1, device="cuda"); eps = torch.rand(B, 1, 1, 1, device="cuda")
with torch.no_grad():
    fake = G(z)
params = list(D.parameters())
res = {}
for fused in (False, True, False, True):
    ops.clear_cast_cache()
    mixed = (real * eps + fake * (1 - eps)).requires_grad_(True)
    scores = D(mixed)
    with ops.input_gradients_only():
        gradient = torch.autograd.grad(scores.sum(), mixed, create_graph=True)[0]
    if fused:
        pen = ops.gradient_penalty(gradient, 10.0)[0]
    else:
        pen = 10.0 * ((gradient.reshape(B, -1).norm(2, dim=1) - 1) ** 2).mean()
    u = torch.autograd.grad(pen, gradient, retain_graph=True)[0]
    g_via_u = torch.autograd.grad(gradient, params, grad_outputs=u, retain_graph=True)
    g_via_ref_u = torch.autograd.grad(gradient, params, grad_outputs=res.get("u", u), retain_graph=True)
    g_via_u2 = torch.autograd.grad(gradient, params, grad_outputs=u, retain_graph=True)
    grabbed = []
    h = gradient.register_hook(lambda t: grabbed.append(t.clone()))
    g = torch.autograd.grad(pen, params, retain_graph=True)
    h.remove()
    g_again = torch.autograd.grad(pen, params)
    print("   u seen by the hook vs u:", (grabbed[0] - u).abs().max().item(), grabbed[0].stride(), " same u twice:",
          " ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(g_via_u2, g_via_u)), " pen->params twice:",
          " ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(g_again, g)))
    print("   pen->params vs gradient->params with own u:", " ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(g, g_via_u)),
          " with the first run's u:", " ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(g, g_via_ref_u)))
    if not res:
        res["u"], res["g"], res["grad"] = u, g, gradient.detach()
    print("fused" if fused else "torch", "pen", pen.item(), "gradient rel", ((gradient.detach() - res["grad"]).norm() / res["grad"].norm()).item(),
          "u rel", ((u - res["u"]).norm() / res["u"].norm()).item(), "u absmax", u.abs().max().item(), "u min nonzero", u[u != 0].abs().min().item(),
          "params:", " ".join(f"{((a - b).norm() / b.norm()).item():.2e}" for a, b in zip(g, res["g"])))
    ub, rb = u.to(torch.bfloat16), res["u"].to(torch.bfloat16)
    print("    bf16(u) differing elements:", (ub != rb).sum().item(), "of", ub.numel())
