"""Developer tool: ms per step of one model family of the benchmark line (gan_amd/harness/steps.py), recorded as one HIP
graph:   python tools/family_step.py pix2pix256_b8 [dcgan64_b128 stylegan256_b32 stylegan256_b32:sweeps]
(":sweeps": the StyleGAN critic with its LeakyReLU derivatives as sweeps of their own, for an A/B on one box)"""
import sys
import time
sys.path.insert(0, ".")
import torch
import gan_amd
from gan_amd.harness import steps

for name in sys.argv[1:] or list(steps.FAMILIES):
    name, _, variant = name.partition(":")
    if variant == "sweeps":
        from gan_amd.harness import stylegan
        stylegan.set_deferred_activations(False)
    step = steps.FAMILIES[name]("cuda:0")
    batch = step.batch
    g = gan_amd.GraphedStep(step)
    for _ in range(3):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name}{':' + variant if variant else ''}: {ms:.3f} ms per step, {batch / ms * 1e3:.0f} img/s", flush=True)
    del g, step
    steps.reset_options()
    if variant == "sweeps":
        stylegan.set_deferred_activations(True)
