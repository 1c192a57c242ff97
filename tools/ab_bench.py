"""Developer tool: A/B timing of the recorded WGAN-GP cycle inside ONE process (box-to-box spread is about 2 %, more than
most single changes): python tools/ab_bench.py [--what two_streams|repack]  -> ms per cycle with and without the option."""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_amd
from gan_amd.harness.basic_gans import WganGpTrainer, synth_inputs

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="two_streams", choices=["two_streams", "three_chains", "repack", "gen_ahead", "flag", "freeze_critic"])
ap.add_argument("--flag", default="weight_gradient_sink", help="with --what flag: the boolean switch of gan_amd.harness.basic_gans under test, or NAME=A,B for two integer values")
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()


def make(on):
    if args.what == "flag":
        from gan_amd.harness import basic_gans
        if "=" in args.flag:          # NAME=A,B: the integer value A against B
            name, vals = args.flag.split("=")
            setattr(basic_gans, name, int(vals.split(",")[0 if on else 1]))
        else:
            setattr(basic_gans, args.flag, bool(on))
    # three_chains: real, fake and penalty passes on three streams against real + fake batched beside the penalty pass
    tr = WganGpTrainer(args.batch, 64, 100, 64, device="cuda:0", capturable=True, batch_real_fake=not (on and args.what == "three_chains"),
                       fused_adam=True, activation_dtype=torch.bfloat16,
                       two_streams=(on and args.what == "two_streams") or args.what in ("three_chains", "gen_ahead", "flag", "freeze_critic"))
    tr.gen_ahead = tr.two_streams and (on or args.what != "gen_ahead")
    tr.freeze_critic_in_generator_phase = on and args.what == "freeze_critic"
    if args.what == "repack" and not on:
        tr.opt_D.repack_after_step()
        tr.opt_G.repack_after_step()
    g = torch.Generator().manual_seed(5001)
    real, zs = synth_inputs(g, args.batch, 100, 64, 6)
    tr.set_inputs(real, zs, torch.rand(5, args.batch, 1, 1, 1, generator=g))
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
    print(f"round {rnd}: {args.what if args.what != 'flag' else args.flag} on {timed(a):.3f} ms   off {timed(b):.3f} ms", flush=True)
print("losses on :", a.static_out.cpu().numpy().round(4).tolist())
print("losses off:", b.static_out.cpu().numpy().round(4).tolist())
