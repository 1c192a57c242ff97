"""Probe: how much does the GPU gain from two independent WGAN-GP cycles in flight at once (two models, two streams)
compared with running them back to back?  Upper bound for what overlapping the independent chains of one critic
iteration could give."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gan_amd.harness.basic_gans import WganGpTrainer, synth_inputs


def make(batch):
    tr = WganGpTrainer(batch, 64, 100, 64, device="cuda:0", capturable=True, batch_real_fake=True, fused_adam=True, activation_dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(5001)
    real, zs = synth_inputs(g, batch, 100, 64, 6)
    tr.set_inputs(real, zs, torch.rand(5, batch, 1, 1, 1, generator=g))
    tr.capture(warmup=2)
    return tr


for batch in (256, 128):
    a, b = make(batch), make(batch)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def run(n, concurrent):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            if concurrent:
                with torch.cuda.stream(s1):
                    a.graph.replay()
                with torch.cuda.stream(s2):
                    b.graph.replay()
            else:
                a.graph.replay(); b.graph.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    run(3, False); run(3, True)
    print(f"batch {batch}: two cycles back to back {run(10, False):.2f} ms, on two streams {run(10, True):.2f} ms", flush=True)
