"""Developer tool: per-parameter relative L2 error of the end-to-end gradients of the Pix2Pix / StyleGAN / BigGAN harnesses
against the reference traces (tests/golden), by parameter class (convolution / linear weights with >= 2 dimensions, vectors),
for each precision mode: the numbers the tests' bounds are set from.   python tools/grad_error_report.py"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import gan_amd as gan
from test_oracle_golden import pix2pix_setup, biggan_setup, stylegan_setup, check_grads, rel_err
from test_gpu_parity import cu
gd = os.path.join("tests", "golden")


def summarise(tag, params, report):
    big = {k: e for k, e in report.items() if params[k].dim() >= 2}
    small = {k: e for k, e in report.items() if params[k].dim() < 2}
    def top(d):
        return ", ".join(f"{k.split('.')[-3] if k.count('.') > 2 else k}:{e:.1e}" for k, e in sorted(d.items(), key=lambda kv: -kv[1])[:3])
    print(f"{tag:60s} weights n={len(big):3d} max {max(big.values(), default=0):.2e} | vectors n={len(small):3d} max {max(small.values(), default=0):.2e} | worst weights: {top(big)} | worst vectors: {top(small)}", flush=True)


for prec, fast in (("fp32x3", False), ("bf16", False), ("bf16", True)):
    gan.set_precision(prec)
    from gan_amd.harness import pix2pix as P, stylegan as S, biggan as Bg
    for fixture in ("pix2pix_trace", "pix2pix_wide_trace"):
        # (full width runs on the library's BatchNorm kernels only: torch's own batch_norm, i.e. MIOpen, crashes on the
        #  [1, 128, 64, 64] channels_last tensor of this image)
        P.set_fused_norm(bool(fast) or fixture == "pix2pix_wide_trace"); P.set_activation_dtype(torch.bfloat16 if fast else None)
        z, rec, G, D, real, label, md, mg = pix2pix_setup(gd, lambda ch: P.Generator(ch=ch), lambda ch: P.Discriminator(ch=ch), fixture)
        G.cuda(); D.cuda(); real, label = real.cuda(), label.cuda(); md = [m.cuda() for m in md]; mg = [m.cuda() for m in mg]
        d_loss = P.discriminator_phase(G, D, real, label, md)
        rep = {}
        for k, p in D.named_parameters():
            want = z["d.gradD." + k]
            g = p.grad.detach().cpu().numpy()
            g = g if (g.size <= 20000 or want.size == g.size) else g.ravel()[::max(1, g.size // 4096)][:4096]
            if np.linalg.norm(want) > 1e-5 * max(np.linalg.norm(z["d.gradD." + kk]) for kk, _ in D.named_parameters()):
                rep[k] = rel_err(g, want)
        summarise(f"{fixture} {prec} fast={fast} D (loss err {abs(d_loss.item() - z['d.loss'][0]):.1e})", dict(D.named_parameters()), rep)
        g_err, fake = P.generator_phase(G, D, real, label, 100.0, mg)
        rep = {}
        biggest = max(z["g.gradG_norm." + k][0] for k, _ in G.named_parameters())
        for k, p in G.named_parameters():
            if z["g.gradG_norm." + k][0] < 1e-5 * biggest: continue
            g = p.grad.detach().cpu().numpy(); g = g if g.size <= 20000 else g.ravel()[::max(1, g.size // 4096)][:4096]
            rep[k] = rel_err(g, z["g.gradG." + k])
        summarise(f"{fixture} {prec} fast={fast} G (loss rel err {abs(g_err.item() - z['g.loss'][0]) / z['g.loss'][0]:.1e})", dict(G.named_parameters()), rep)
        P.set_fused_norm(False); P.set_activation_dtype(None)
        del G, D
    for fixture in ("biggan_trace", "biggan_wide_trace"):
        Bg.set_activation_dtype(torch.bfloat16 if fast else None); Bg.set_fused_attention(fast)
        z, rec, G, D, real, z1, z2, labels = biggan_setup(gd, Bg, fixture)
        G.cuda(); D.cuda(); real, z1, z2, labels = real.cuda(), z1.cuda(), z2.cuda(), labels.cuda()
        Bg.discriminator_phase(G, D, real, labels[0], z1, labels[1])
        rep = {}; check_grads(D.named_parameters(), z, "d.gradD", 10.0, rep); summarise(f"{fixture} {prec} fast={fast} D", dict(D.named_parameters()), rep)
        Bg.generator_phase(G, D, z2, labels[2])
        rep = {}; check_grads(G.named_parameters(), z, "g.gradG", 10.0, rep); summarise(f"{fixture} {prec} fast={fast} G", dict(G.named_parameters()), rep)
        Bg.set_activation_dtype(None); Bg.set_fused_attention(False)
        del G, D
    for fixture in ("stylegan_trace", "stylegan256_trace"):
        gan.set_split_reduction(False)
        S.set_activation_dtype(torch.bfloat16 if fast else None)
        z, rec, G, D, real, zs, noises = stylegan_setup(gd, lambda: S.StyleBasedGenerator(fused=True), lambda: S.Discriminator(fused=True, use_activations=True), fixture)
        G.cuda(); D.cuda(); real, zs = real.cuda(), zs.cuda(); noises = {k: [t.cuda() for t in v] for k, v in noises.items()}
        pg = rec["pg_step"]
        S.critic_phase(G, D, real, [zs[0], zs[1]], cu(z["critic.eps"]), pg, 1, 10.0, noises["critic"], int(z["critic.crossover"][0]))
        rep = {}; check_grads(D.named_parameters(), z, "critic.gradD", 10.0, rep); summarise(f"{fixture} {prec} fast={fast} D", dict(D.named_parameters()), rep)
        S.generator_phase(G, D, [zs[2], zs[3]], pg, 1, noises["gen"], int(z["gen.crossover"][0]))
        rep = {}; check_grads(G.named_parameters(), z, "gen.gradG", 10.0, rep); summarise(f"{fixture} {prec} fast={fast} G", dict(G.named_parameters()), rep)
        S.set_activation_dtype(None); gan.set_split_reduction(True)
        del G, D
    torch.cuda.empty_cache()
