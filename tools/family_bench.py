"""Developer tool: one training step (discriminator phase + generator phase, gradients only) of each of the
reference's model families on the GPU harnesses, against the CPU restatement of the reference (oracle/) on the host
cores, at the reference's default widths.  Prints a table; run on the GPU box:
    python tools/family_bench.py            reduced sizes (seconds)
    python tools/family_bench.py --full     per-GPU sizes of the reference's configurations (Pix2Pix 256x256 batch 8,
                                            StyleGAN 256x256 batch 32, BigGAN 128x128 ch 64 batch 256); the CPU side
                                            runs batch 4 and its rate is taken as independent of the batch"""
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
FULL = "--full" in sys.argv
CPU_B = 4
ONLY = [a.split("=")[1] for a in sys.argv if a.startswith("--only=")]     # --only=biggan etc.
NO_CPU = "--no-cpu" in sys.argv
GRAPH = "--graph" in sys.argv    # also time the step recorded into one HIP graph
TOP = ([int(a.split("=")[1]) for a in sys.argv if a.startswith("--top=")] or [10])[0]      # kernels listed per family
FAST = "--fast" in sys.argv      # bf16 activation storage and the other speed options of the harnesses


def want(name):
    return not ONLY or name in ONLY


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


def graph_time(step, n=5):
    if not GRAPH:
        return float("nan")
    g = gan_amd.GraphedStep(lambda: (step(), ops.clear_cast_cache()) and None)
    for _ in range(2):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


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
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:TOP]:
        tf = v["flops"] / (v["total_ms"] * 1e-3) / 1e12 if v["flops"] > 0 and v["total_ms"] > 0 else 0.0
        print(f"   {k:78s} n={v['launches']:4d} {v['total_ms']:8.3f} ms {tf:7.1f} TFLOP/s")


def cpu_time(step):
    if NO_CPU:
        return float("nan")
    t0 = time.perf_counter()
    step()
    return time.perf_counter() - t0


def note(msg):
    print(msg, flush=True)


def zero(*nets):
    for n in nets:
        for p in n.parameters():
            p.grad = None


rows = []

if FAST:
    pix2pix.set_activation_dtype(torch.bfloat16)
    pix2pix.set_fused_norm(True)

if want("pix2pix"):
    # ---- Pix2Pix: U-Net (ch 64) + PatchGAN on 256x256, batch 4
    B = 8 if FULL else 4
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
        ref_models.pix2pix_discriminator_phase(Gr, Dr, real[:CPU_B], label[:CPU_B])
        ref_models.pix2pix_generator_phase(Gr, Dr, real[:CPU_B], label[:CPU_B])


    rows.append((f"Pix2Pix U-Net + PatchGAN, 256x256, batch {B}", B, gpu_time(p2p_gpu), cpu_time(p2p_cpu), graph_time(p2p_gpu)))
    note("pix2pix done")
    kernel_table(p2p_gpu, "Pix2Pix")
    del G, D, Gr, Dr

if want("dcgan"):
    # ---- DCGAN 64x64 (hidden 64, nz 100), batch 128: one discriminator + one generator update (BasicGANs/DCGAN.py:165-181)
    B = 128
    act = torch.bfloat16 if FAST else torch.float32
    torch.manual_seed(5000)
    G = basic_gans.Generator(100, 64, 3, 64, activation_dtype=act).to(dev)
    D = basic_gans.Discriminator(3, 64, 64, is_wgan=False, activation_dtype=act).to(dev)
    G.apply(basic_gans.weights_init); D.apply(basic_gans.weights_init)
    Gr, Dr = ref_models.Generator(100, 64, 3, resolution=64), ref_models.Discriminator(3, 64, resolution=64)
    Gr.apply(ref_models.weights_init); Dr.apply(ref_models.weights_init)
    oG = gan_amd.optim.Adam(G.parameters(), lr=2e-4, betas=(0.5, 0.999)); oD = gan_amd.optim.Adam(D.parameters(), lr=2e-4, betas=(0.5, 0.999))
    oGr = torch.optim.Adam(Gr.parameters(), lr=2e-4, betas=(0.5, 0.999)); oDr = torch.optim.Adam(Dr.parameters(), lr=2e-4, betas=(0.5, 0.999))
    real, z = torch.rand(B, 3, 64, 64) * 2 - 1, torch.randn(B, 100, 1, 1)
    rg, zg = real.to(dev).contiguous(memory_format=torch.channels_last), z.to(dev)


    def dc_gpu():
        oD.zero_grad(set_to_none=False)
        *_, fake = basic_gans.dcgan_discriminator_phase(G, D, rg, zg)
        oD.step()
        oG.zero_grad(set_to_none=False)
        basic_gans.dcgan_generator_phase(D, fake)
        oG.step()


    def dc_cpu():
        for _ in range(4):     # four steps, scaled below: one step is too short to time
            ref_models.dcgan_train_step(Gr, Dr, oGr, oDr, real, z)


    t_cpu = cpu_time(dc_cpu) / 4 * CPU_B / B       # (the table divides CPU_B by the time: rescale to the full batch)
    rows.append((f"DCGAN 64x64, batch {B}, with both Adam steps", B, gpu_time(dc_gpu), t_cpu, graph_time(dc_gpu)))
    note("dcgan done")
    kernel_table(dc_gpu, "DCGAN")
    del G, D, Gr, Dr

if want("pix2pixhd"):
    # ---- Pix2PixHD global generator (ch 64, 9 residual blocks) + PatchGAN on 256x256 (Pix2Pix/train.py:27 selects it by name)
    B = 8 if FULL else 4
    real, label = torch.rand(B, 3, 256, 256) * 2 - 1, torch.rand(B, 3, 256, 256)
    G, D = pix2pix.GeneratorHD(ch=64).to(dev), pix2pix.Discriminator(ch=64).to(dev)
    Gr, Dr = ref_models.P2PGeneratorHD(ch=64), ref_models.P2PDiscriminator(ch=64)
    rg, lg = real.to(dev), label.to(dev)


    def hd_gpu():
        zero(G, D)
        pix2pix.discriminator_phase(G, D, rg, lg)
        pix2pix.generator_phase(G, D, rg, lg)


    def hd_cpu():
        zero(Gr, Dr)
        ref_models.pix2pix_discriminator_phase(Gr, Dr, real[:CPU_B], label[:CPU_B])
        ref_models.pix2pix_generator_phase(Gr, Dr, real[:CPU_B], label[:CPU_B])


    rows.append((f"Pix2PixHD global generator + PatchGAN, 256x256, batch {B}", B, gpu_time(hd_gpu), cpu_time(hd_cpu), graph_time(hd_gpu)))
    note("pix2pixhd done")
    kernel_table(hd_gpu, "Pix2PixHD")
    del G, D, Gr, Dr

if want("stylegan"):
    # ---- StyleGAN-v1 at progressive step 5 (128x128), batch 16, style mixing, gradient penalty
    if FAST:
        stylegan.set_activation_dtype(torch.bfloat16)
    B, pg = (32, 6) if FULL else (16, 5)
    real = torch.rand(B, 3, 4 * 2 ** pg, 4 * 2 ** pg) * 2 - 1
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
        ref_stylegan.critic_phase(Gr, Dr, real[:CPU_B], [zs[0, :CPU_B], zs[1, :CPU_B]], eps[:CPU_B], pg, crossover_point=3)
        ref_stylegan.generator_phase(Gr, Dr, [zs[2, :CPU_B], zs[3, :CPU_B]], pg, crossover_point=3)


    rows.append((f"StyleGAN-v1, progressive step {pg} ({4 * 2 ** pg}x{4 * 2 ** pg}), batch {B}", B, gpu_time(sg_gpu, 3), cpu_time(sg_cpu), graph_time(sg_gpu, 3)))
    note("stylegan done")
    kernel_table(sg_gpu, "StyleGAN")
    del G, D, Gr, Dr

if want("biggan"):
    # ---- BigGAN 128x128 (G ch 96, D ch 64), 10 classes, batch 16
    if FAST:
        biggan.set_activation_dtype(torch.bfloat16)
        biggan.set_fused_attention(True)
    B, ncls = (256 if FULL else 16), 10
    gch = 64 if FULL else 96
    real = torch.rand(B, 3, 128, 128) * 2 - 1
    z1, z2 = torch.rand(B, 120), torch.rand(B, 120)
    lab = torch.randint(0, ncls, (3, B))
    G, D = biggan.Generator(n_classes=ncls, ch=gch).to(dev), biggan.Discriminator(n_classes=ncls, ch=64).to(dev)
    Gr, Dr = ref_biggan.Generator(n_classes=ncls, ch=gch), ref_biggan.Discriminator(n_classes=ncls, ch=64)
    rg, z1g, z2g, labg = real.to(dev), z1.to(dev), z2.to(dev), lab.to(dev)


    def bg_gpu():
        zero(G, D)
        biggan.discriminator_phase(G, D, rg, labg[0], z1g, labg[1])
        biggan.generator_phase(G, D, z2g, labg[2])


    def bg_cpu():
        zero(Gr, Dr)
        ref_biggan.discriminator_phase(Gr, Dr, real[:CPU_B], lab[0, :CPU_B], z1[:CPU_B], lab[1, :CPU_B])
        ref_biggan.generator_phase(Gr, Dr, z2[:CPU_B], lab[2, :CPU_B])


    rows.append((f"BigGAN 128x128 (G ch {gch}, D ch 64), batch {B}", B, gpu_time(bg_gpu, 3), cpu_time(bg_cpu), graph_time(bg_gpu, 3)))
    kernel_table(bg_gpu, "BigGAN")

print(f"{'family (D phase + G phase, gradients only, eager launches)':62s} {'GPU ms':>9s} {'GPU img/s':>10s} {'CPU s':>8s} {'CPU img/s':>10s} {'ratio':>7s}")
for name, b, tg, tc, tgr in rows:
    print(f"{name:62s} {1e3 * tg:9.2f} {b / tg:10.1f} {tc:8.2f} {CPU_B / tc:10.2f} {(b / tg) / (CPU_B / tc):7.1f}" +
          (f"   as one HIP graph: {1e3 * tgr:8.2f} ms {b / tgr:9.1f} img/s" if GRAPH else ""))
print("harness speed options (bf16 activation storage, fused attention):", "on" if FAST else "off")
print(f"(CPU steps run at batch {CPU_B}; torch CPU threads: {torch.get_num_threads()}; bf16 MFMA precision mode on the GPU)")
