"""Developer tool: A/B timing of the recorded WGAN-GP cycle inside ONE process (box-to-box spread is about 2 %, more than
most single changes): python tools/ab_bench.py  -> ms per cycle with and without the multi-weight re-pack."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_amd
from gan_amd.harness.basic_gans import WganGpTrainer, synth_inputs


def make(repack):
    tr = WganGpTrainer(256, 64, 100, 64, device="cuda:0", capturable=True, batch_real_fake=True, fused_adam=True, activation_dtype=torch.bfloat16)
    if not repack:
        tr.opt_D.repack_after_step()
        tr.opt_G.repack_after_step()
    g = torch.Generator().manual_seed(5001)
    real, zs = synth_inputs(g, 256, 100, 64, 6)
    tr.set_inputs(real, zs, torch.rand(5, 256, 1, 1, 1, generator=g))
    tr.capture(warmup=2)
    return tr


def timed(tr, n=20):
    for _ in range(3):
        tr.cycle()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.cycle()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


a, b = make(True), make(False)
for rnd in range(3):
    print(f"round {rnd}: multi-weight re-pack {timed(a):.3f} ms   per-layer re-pack {timed(b):.3f} ms", flush=True)
