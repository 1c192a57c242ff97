"""Developer tool: one training step (discriminator phase + generator phase, gradients only) of each of the
reference's model families on the GPU harnesses, against the CPU restatement of the reference (oracle/) on the host
cores, at the reference's default widths.  Prints a table; run on the GPU box:  python tools/family_bench.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import gan_amd
from gan_amd import ops
from gan_amd.harness import basic_gans, biggan, pix2pix, stylegan
from oracle import ref_biggan, ref_models, ref_stylegan

torch.manual_seed(0)
dev = "cuda:0"


def gpu_time(step, n=5):
    for _ in range(2):
        step(); ops.clear_cast_cache()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        step(); ops.clear_cast_cache()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def kernel_table(step, title):
    """Library kernels of one step (HIP events around each launch)."""
    from gan_amd import _lib
    _lib.profile_enable(True)
    step(); ops.clear_cast_cache()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    tot = sum(v["total_ms"] for v in prof.values())
    print(f"-- {title}: library kernels {tot:.2f} ms per step")
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:10]:
        tf = v["flops"] / (v["total_ms"] * 1e-3) / 1e12 if v["flops"] > 0 and v["total_ms"] > 0 else 0.0
        print(f"   {k:78s} n={v['launches']:4d} {v['total_ms']:8.3f} ms {tf:7.1f} TFLOP/s")


def cpu_time(step):
    t0 = time.perf_counter()
    step()
    return time.perf_counter() - t0


def zero(*nets):
    for n in nets:
        for p in n.parameters():
            p.grad = None


rows = []

# ---- Pix2Pix: U-Net (ch 64) + PatchGAN on 256x256, batch 4
B = 4
real, label = torch.rand(B, 3, 256, 256) * 2 - 1, torch.rand(B, 3, 256, 256)
G, D = pix2pix.Generator(ch=64).to(dev), pix2pix.Discriminator(ch=64).to(dev)
Gr, Dr = ref_models.P2PGenerator(ch=64), ref_models.P2PDiscriminator(ch=64)
rg, lg = real.to(dev), label.to(dev)


def p2p_gpu():
    zero(G, D)
    pix2pix.discriminator_phase(G, D, rg, lg)
    pix2pix.generator_phase(G, D, rg, lg)


def p2p_cpu():
    zero(Gr, Dr)
    ref_models.pix2pix_discriminator_phase(Gr, Dr, real, label)
    ref_models.pix2pix_generator_phase(Gr, Dr, real, label)


rows.append(("Pix2Pix U-Net + PatchGAN, 256x256, batch 4", B, gpu_time(p2p_gpu), cpu_time(p2p_cpu)))
kernel_table(p2p_gpu, "Pix2Pix")
del G, D, Gr, Dr

# ---- StyleGAN-v1 at progressive step 5 (128x128), batch 16, style mixing, gradient penalty
B, pg = 16, 5
real = torch.rand(B, 3, 128, 128) * 2 - 1
zs = torch.randn(4, B, 512)
eps = torch.rand(B, 1, 1, 1)
G, D = stylegan.StyleBasedGenerator().to(dev), stylegan.Discriminator().to(dev)
Gr, Dr = ref_stylegan.StyleBasedGenerator(), ref_stylegan.Discriminator()
rg, zg, eg = real.to(dev), zs.to(dev), eps.to(dev)


def sg_gpu():
    zero(G, D)
    stylegan.critic_phase(G, D, rg, [zg[0], zg[1]], eg, pg, crossover_point=3)
    stylegan.generator_phase(G, D, [zg[2], zg[3]], pg, crossover_point=3)


def sg_cpu():
    zero(Gr, Dr)
    ref_stylegan.critic_phase(Gr, Dr, real, [zs[0], zs[1]], eps, pg, crossover_point=3)
    ref_stylegan.generator_phase(Gr, Dr, [zs[2], zs[3]], pg, crossover_point=3)


rows.append(("StyleGAN-v1, progressive step 5 (128x128), batch 16", B, gpu_time(sg_gpu), cpu_time(sg_cpu)))
kernel_table(sg_gpu, "StyleGAN")
del G, D, Gr, Dr

# ---- BigGAN 128x128 (G ch 96, D ch 64), 10 classes, batch 16
B, ncls = 16, 10
real = torch.rand(B, 3, 128, 128) * 2 - 1
z1, z2 = torch.rand(B, 120), torch.rand(B, 120)
lab = torch.randint(0, ncls, (3, B))
G, D = biggan.Generator(n_classes=ncls, ch=96).to(dev), biggan.Discriminator(n_classes=ncls, ch=64).to(dev)
Gr, Dr = ref_biggan.Generator(n_classes=ncls, ch=96), ref_biggan.Discriminator(n_classes=ncls, ch=64)
rg, z1g, z2g, labg = real.to(dev), z1.to(dev), z2.to(dev), lab.to(dev)


def bg_gpu():
    zero(G, D)
    biggan.discriminator_phase(G, D, rg, labg[0], z1g, labg[1])
    biggan.generator_phase(G, D, z2g, labg[2])


def bg_cpu():
    zero(Gr, Dr)
    ref_biggan.discriminator_phase(Gr, Dr, real, lab[0], z1, lab[1])
    ref_biggan.generator_phase(Gr, Dr, z2, lab[2])


rows.append(("BigGAN 128x128 (G ch 96, D ch 64), batch 16", B, gpu_time(bg_gpu), cpu_time(bg_cpu)))
kernel_table(bg_gpu, "BigGAN")

print(f"{'family (D phase + G phase, gradients only, eager launches)':62s} {'GPU ms':>9s} {'GPU img/s':>10s} {'CPU s':>8s} {'CPU img/s':>10s} {'ratio':>7s}")
for name, b, tg, tc in rows:
    print(f"{name:62s} {1e3 * tg:9.2f} {b / tg:10.1f} {tc:8.2f} {b / tc:10.2f} {tc / tg:7.1f}")
print(f"(torch CPU threads: {torch.get_num_threads()}; bf16 MFMA precision mode on the GPU)")
