"""Probe: Pix2Pix step, eager twice vs graphed, per step losses (developer aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan_amd as gan
from gan_amd.harness import pix2pix as P

B, ch = 2, 8
MODE = sys.argv[1] if len(sys.argv) > 1 else "plain"
gen = torch.Generator().manual_seed(11)
batches = [(torch.rand(B, 3, 256, 256, generator=gen) * 2 - 1, torch.rand(B, 3, 256, 256, generator=gen)) for _ in range(5)]
masks = [(torch.rand(B, ch * 8, r, r, generator=gen) < 0.5).float().cuda() * 2 for r in (2, 4, 8)]


def make():
    torch.manual_seed(12)
    G, D = P.Generator(ch=ch).cuda(), P.Discriminator(ch=ch).cuda()
    A = torch.optim.Adam if "torchadam" in MODE else gan.optim.Adam
    kw = dict(capturable=True) if "torchadam" in MODE else {}
    oG = A(G.parameters(), lr=2e-4, betas=(0.5, 0.999), **kw); oD = A(D.parameters(), lr=2e-4, betas=(0.5, 0.999), **kw)
    if "torchD" in MODE:
        oD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    if "torchG" in MODE:
        oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999), capturable=True)
    real, label = torch.empty(B, 3, 256, 256, device="cuda"), torch.empty(B, 3, 256, 256, device="cuda")

    def step():
        oD.zero_grad(set_to_none=False); oG.zero_grad(set_to_none=False)
        d = P.discriminator_phase(G, D, real, label, masks)
        oD.step()
        g, _ = P.generator_phase(G, D, real, label, 100.0, masks)
        oG.step()
        if "clear" in MODE:
            gan.ops.clear_cast_cache()
        return torch.stack([d, g])

    def feed(i):
        real.copy_(batches[i][0]); label.copy_(batches[i][1])
    step.nets = (G, D, oG, oD)
    return step, feed


def chk(step):
    G, D, oG, oD = step.nets
    f = lambda ps: float(sum(p.detach().double().abs().sum() for p in ps))
    mom = lambda o: float(sum(st["exp_avg_sq"].double().sum() for st in o.state.values() if "exp_avg_sq" in st))
    gr = lambda ps: float(sum(p.grad.double().abs().sum() for p in ps if p.grad is not None))
    return "G %.6f D %.6f vG %.6e vD %.6e gG %.5e gD %.5e" % (f(G.parameters()), f(D.parameters()), mom(oG), mom(oD), gr(G.parameters()), gr(D.parameters()))


gan.set_split_reduction(False)
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
if "ref" in mode:
    step, feed = make()
    for i in range(5):
        feed(i); step()
step, feed = make()
feed(0)
graphed = gan.GraphedStep(step, warmup=2)
step_e, feed_e = make()
feed_e(0)
for _ in range(2):
    step_e()
for i in range(1, 5):
    feed(i); out = graphed().cpu().clone()
    if "alloc" in mode:
        junk = [torch.randn(1 << 20, device="cuda") for _ in range(64)]
        junk = [j * 2 for j in junk]
        del junk
        print(mode, i, "graph", out.tolist(), flush=True)
    elif "inter" in mode:
        feed_e(i); want = step_e().cpu()
        print(mode, i, "graph", out.tolist(), "eager", want.tolist(), flush=True)
        print("   graph", chk(step)); print("   eager", chk(step_e), flush=True)
        if i == 2:
            Dg, De = step.nets[1], step_e.nets[1]
            for (k, a), (_, b) in zip(Dg.named_parameters(), De.named_parameters()):
                print("      %-22s grad ratio %.4f  |g| graph %.4e eager %.4e   param diff %.3e" % (
                    k, float(a.grad.norm() / b.grad.norm()), float(a.grad.norm()), float(b.grad.norm()), float((a - b).abs().max())), flush=True)
    else:
        print(mode, i, "graph", out.tolist(), flush=True)
