"""Developer tool: per-parameter gradient errors of the StyleGAN harness against the golden trace (GPU box).
   python tools/stylegan_grad_errors.py [bf16|fp32x3] [--bf16-storage]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gan_amd as gan
from gan_amd.harness import stylegan as S
from test_oracle_golden import check_grads, stylegan_setup

golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
gan.set_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16")
S.set_activation_dtype(torch.bfloat16 if "--bf16-storage" in sys.argv else None)
z, rec, G, D, real, zs, noises = stylegan_setup(golden, lambda: S.StyleBasedGenerator(fused=True), lambda: S.Discriminator(fused=True, use_activations=True))
G.cuda(); D.cuda()
real, zs = real.cuda(), zs.cuda()
noises = {k: [t.cuda() for t in v] for k, v in noises.items()}
pg = rec["pg_step"]
d_x, d_g, gp, _ = S.critic_phase(G, D, real, [zs[0], zs[1]], torch.from_numpy(z["critic.eps"]).cuda(), pg, 1, 10.0, noises["critic"], int(z["critic.crossover"][0]))
print("critic scalars", d_x.item(), d_g.item(), gp.item(), "want", z["critic.scalars"])
rep = {}
check_grads(D.named_parameters(), z, "critic.gradD", 10.0, rep)
print("D worst:", sorted(rep.items(), key=lambda kv: -kv[1])[:5])
g_err, _ = S.generator_phase(G, D, [zs[2], zs[3]], pg, 1, noises["gen"], int(z["gen.crossover"][0]))
print("g_err", g_err.item(), "want", z["gen.scalars"][0])
rep = {}
check_grads(G.named_parameters(), z, "gen.gradG", 10.0, rep)
for k, v in sorted(rep.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v:8.4f}  {k}")
