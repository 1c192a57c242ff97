import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import gan_amd as gan
from oracle import bf16_operands as E
from oracle import ref_models as R
from test_oracle_golden import biggan_setup, pix2pix_setup, rel_err, stylegan_setup
gd = "tests/golden"
gan.set_precision("bf16")

def errs(named_gpu, named_cpu, tag):
    cpu = dict(named_cpu)
    biggest = max(float(p.grad.norm()) for p in cpu.values() if p.grad is not None)
    out = []
    for k, p in named_gpu:
        q = cpu[k]
        if q.grad is None or p.grad is None: continue
        w = q.grad.numpy(); g = p.grad.detach().float().cpu().numpy()
        if np.linalg.norm(w) < 1e-5 * biggest: continue
        out.append((rel_err(g, w), k))
    out.sort(reverse=True)
    print(tag, "n", len(out), "worst", [(f"{e:.1e}", k) for e, k in out[:6]], "median", f"{out[len(out)//2][0]:.1e}", flush=True)

from gan_amd.harness import pix2pix as P
z, rec, G, D, real, label, md, mg = pix2pix_setup(gd, lambda ch: P.Generator(ch=ch), lambda ch: P.Discriminator(ch=ch))
_, _, Gr, Dr, _, _, _, _ = pix2pix_setup(gd, lambda ch: R.P2PGenerator(ch=ch), lambda ch: R.P2PDiscriminator(ch=ch))
with E.conv_operands_rounded():
    d_ref = R.pix2pix_discriminator_phase(Gr, Dr, real, label, md)
G.cuda(); D.cuda()
d = P.discriminator_phase(G, D, real.cuda(), label.cuda(), [m.cuda() for m in md])
print("p2p d loss", d.item(), d_ref.item())
errs(D.named_parameters(), Dr.named_parameters(), "p2p D")
with E.conv_operands_rounded():
    g_ref, fake_ref = R.pix2pix_generator_phase(Gr, Dr, real, label, 100.0, mg)
g, fake = P.generator_phase(G, D, real.cuda(), label.cuda(), 100.0, [m.cuda() for m in mg])
print("p2p g loss", g.item(), g_ref.item(), "fake err", rel_err(fake.cpu().numpy(), fake_ref.numpy()))
errs(G.named_parameters(), Gr.named_parameters(), "p2p G")

from gan_amd.harness import biggan as Bg
from oracle import ref_biggan as Rb
z, rec, G, D, real, z1, z2, labels = biggan_setup(gd, Bg)
_, _, Gr, Dr, _, _, _, _ = biggan_setup(gd, Rb)
with E.conv_operands_rounded():
    dr = Rb.discriminator_phase(Gr, Dr, real, labels[0], z1, labels[1])
G.cuda(); D.cuda()
d = Bg.discriminator_phase(G, D, real.cuda(), labels[0].cuda(), z1.cuda(), labels[1].cuda())
print("biggan d", [t.item() for t in d], [t.item() for t in dr])
errs(D.named_parameters(), Dr.named_parameters(), "biggan D")
with E.conv_operands_rounded():
    gr = Rb.generator_phase(Gr, Dr, z2, labels[2])
g = Bg.generator_phase(G, D, z2.cuda(), labels[2].cuda())
print("biggan g", g.item(), gr.item())
errs(G.named_parameters(), Gr.named_parameters(), "biggan G")

from gan_amd.harness import stylegan as S
from oracle import ref_stylegan as Rs
gan.set_split_reduction(False)
z, rec, G, D, real, zs, noises = stylegan_setup(gd, lambda: S.StyleBasedGenerator(fused=True), lambda: S.Discriminator(fused=True, use_activations=True))
_, _, Gr, Dr, _, _, _ = stylegan_setup(gd, lambda: Rs.StyleBasedGenerator(fused=True), lambda: Rs.Discriminator(fused=True, use_activations=True))
pg, eps = rec["pg_step"], torch.from_numpy(z["critic.eps"])
cc, cg = int(z["critic.crossover"][0]), int(z["gen.crossover"][0])
with E.conv_operands_rounded():
    ref = Rs.critic_phase(Gr, Dr, real, [zs[0], zs[1]], eps, pg, 1, 10.0, noises["critic"], cc)
G.cuda(); D.cuda()
cn = {k: [t.cuda() for t in v] for k, v in noises.items()}
got = S.critic_phase(G, D, real.cuda(), [zs[0].cuda(), zs[1].cuda()], eps.cuda(), pg, 1, 10.0, cn["critic"], cc)
print("sg critic", [t.item() for t in got[:3]], [t.item() for t in ref[:3]], "fake err", rel_err(got[3].float().cpu().numpy(), ref[3].numpy()))
errs(D.named_parameters(), Dr.named_parameters(), "sg D")
with E.conv_operands_rounded():
    gr = Rs.generator_phase(Gr, Dr, [zs[2], zs[3]], pg, 1, noises["gen"], cg)
g = S.generator_phase(G, D, [zs[2].cuda(), zs[3].cuda()], pg, 1, cn["gen"], cg)
print("sg gen", g[0].item(), gr[0].item())
errs(G.named_parameters(), Gr.named_parameters(), "sg G")
