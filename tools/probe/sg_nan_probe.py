"""Probe: which backward pass of the StyleGAN critic phase produces non-finite gradients under bf16 storage."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import gan_amd as gan
from gan_amd.harness import stylegan as S
from test_oracle_golden import stylegan_setup

gan.set_precision("bf16")
S.set_activation_dtype(torch.bfloat16)
z, rec, G, D, real, zs, noises = stylegan_setup(os.path.join(root, "tests", "golden"), lambda: S.StyleBasedGenerator(fused=True), lambda: S.Discriminator(fused=True, use_activations=True))
G.cuda(); D.cuda(); real = real.cuda(); zs = zs.cuda()
pg = rec["pg_step"]
kw = dict(progressive_growing_phase=pg, alpha=1)


def bad(tag):
    names = [n for n, p in D.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(tag, "non-finite grads:", names[:6], flush=True)
    for p in D.parameters():
        p.grad = None


for m in D.modules():
    if not list(m.children()):
        m.register_forward_hook(lambda mod, i, o: print("   non-finite output of", type(mod).__name__) if not torch.isfinite(o.float()).all() else None)
d_real = D(real, **kw).view(-1)
(-d_real.mean()).backward(); bad("real pass")
with torch.no_grad():
    fake = G([zs[0], zs[1]], crossover_point=3, **kw)
print("fake finite", torch.isfinite(fake).all().item(), fake.dtype)
D(fake, **kw).mean().backward(); bad("fake pass")
torch.autograd.set_detect_anomaly(True)
eps = torch.from_numpy(z["critic.eps"]).cuda().requires_grad_(True)
mixed = real * eps + fake * (1 - eps)
scores = D(mixed, **kw)
g = torch.autograd.grad(scores.sum(), mixed, create_graph=True)[0]
print("gradient finite", torch.isfinite(g).all().item(), g.dtype)
pen = ((g.reshape(g.shape[0], -1).norm(2, dim=1) - 1) ** 2).mean()
try:
    pen.backward()
except RuntimeError as e:
    print("anomaly:", str(e)[:300])
bad("penalty pass")
