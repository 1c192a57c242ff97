"""Benchmark of the convolution hot path inside the reference's WGAN-GP-64 training cycle on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one training cycle of BasicGANs/train.py:95-119 on synthetic data: 5 critic iterations (each: real
pass, fake pass, gradient-penalty double backward, Adam step) + 1 generator iteration, batch 256 per GPU, i.e.
256 real images consumed per GPU per step.  Prints ONE JSON line (rank 0):
  value        whole-job images/s = n_gpus * 256 * K / (max over ranks of the time for K steps)
  roofline     the dominant conv kernel family: algorithmic FLOPs (2*MACs, necessary taps only) of its launches
               divided by their HIP-event time, measured on the launch stream in a profiled pass of the same
               cycle right after the timed region; peak = dense bf16 MFMA rate of MI355X (2.5 PFLOP/s);
               traffic = HBM bytes per launch of that kernel recorded by separate rocprofv3 counter passes
  cpu_baseline the same cycle on the host CPU through the oracle's restatement of the reference (stock torch
               CPU ops), rank 0 and N=1 only
  families     (N=1) images/s of one TRAINING step (both phases, both Adam steps, the generator's moving average) of the other
               model families at the reference's per-GPU sizes, each replayed as one HIP graph: DCGAN-64 batch 128, Pix2Pix-256
               batch 8 and 64, StyleGAN-256 batch 32, BigGAN-128 batch 256; not part of `value`

    python bench.py --family stylegan256_b32 [--gpus N]      that family's training step as the benchmarked step
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH, HIDDEN, NZ, RES, REPS = 256, 64, 100, 64, 5
PEAK_BF16_TFLOPS = 2500.0


def cpu_baseline(threads):
    """A few full WGAN-GP cycles (batch 256) on the host through oracle.ref_models; returns images/s."""
    from oracle import ref_models as R
    torch.set_num_threads(threads)
    torch.manual_seed(5000)
    G = R.Generator(NZ, HIDDEN, 3, resolution=RES); G.apply(R.weights_init)
    D = R.Discriminator(3, HIDDEN, resolution=RES, is_wgan=True); D.apply(R.weights_init)
    opt_D = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.0, 0.9))
    opt_G = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.0, 0.9))
    gen = torch.Generator().manual_seed(5001)
    real, zs = R.synth_inputs(gen, BATCH, NZ, RES, REPS + 1)
    eps = torch.rand(REPS, BATCH, 1, 1, 1, generator=gen)
    # warm-up: one critic iteration on a quarter batch
    R.wasserstein_step(G, D, real[:64], zs[0][:64], epsilon=eps[0][:64]); opt_D.zero_grad()
    cycles = 4                       # about 12 s of CPU work on 16 cores
    t0 = time.perf_counter()
    for _ in range(cycles):
        R.wgangp_train_cycle(G, D, opt_G, opt_D, real, zs, eps)
    dt = time.perf_counter() - t0
    return {"value": cycles * BATCH / dt, "unit": "img/s", "cores": threads, "kind": "port",
            "sample": f"{cycles} full cycles (5 critic + 1 generator iterations each) at batch 256, oracle/ref_models.py on torch CPU ops, fp32",
            "seconds": dt}


def _dominant(prof, dt):
    """Dominant conv kernel of a profiled eager step (library event timer) and the step's algorithmic conv FLOP rate."""
    conv = {k: v for k, v in prof.items() if v["flops"] > 0}
    if not conv:
        return {}
    dom = max(conv, key=lambda k: conv[k]["total_ms"])
    tf = conv[dom]["flops"] / (conv[dom]["total_ms"] * 1e-3) / 1e12
    flops = sum(v["flops"] for v in conv.values())
    return {"dominant_kernel": dom, "dominant_kernel_tflops": tf, "frac": tf / PEAK_BF16_TFLOPS, "dominant_kernel_launches": conv[dom]["launches"],
            "dominant_kernel_avg_us": 1e3 * conv[dom]["total_ms"] / conv[dom]["launches"],
            "conv_gflop_per_step": flops / 1e9, "step_tflops": flops / dt / 1e12,
            "library_kernel_ms_per_step": sum(v["total_ms"] for v in prof.values())}


def _family_traffic(name, dt):
    """HBM bytes per step of a family from separate rocprofv3 --pmc passes (tools/family_pmc.sh; FETCH_SIZE doubled as the
    gfx950 guide prescribes, summed over every kernel of an eager step), divided by the step time measured here."""
    path = os.path.join(ROOT, "bench_family_traffic.json")
    if not os.path.exists(path):
        return None
    b = json.load(open(path)).get(name, {}).get("hbm_bytes_per_step")
    return None if b is None else b / dt / 1e9


def families(device, steps=8, only=None):
    """{name: {...}} for the training steps of gan_amd.harness.steps (both phases, both Adam steps, the generator's moving
    average), each replayed as one HIP graph; then one eager pass under the library's event timer."""
    import gan_amd
    from gan_amd import _lib
    from gan_amd.harness import steps as S
    out = {}
    for name, make in S.FAMILIES.items():
        if only is not None and name not in only:
            continue
        try:
            step = make(device)
            g = gan_amd.GraphedStep(step)
            for _ in range(2):
                g()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            losses = step.losses.detach().float().cpu().tolist()
            out[name] = {"img_per_s": step.batch / dt, "ms_per_step": 1e3 * dt, "batch": step.batch, "what": step.what, "hip_graph": True,
                         "optimizer_steps": True, "losses": losses, "losses_finite": all(x == x and abs(x) != float("inf") for x in losses)}
            _lib.profile_enable(True)
            step()
            torch.cuda.synchronize()
            prof = _lib.profile_collect()
            _lib.profile_enable(False)
            out[name].update(_dominant(prof, dt))
            out[name]["hbm_gb_s"] = _family_traffic(name, dt)
            del g, step
        except Exception as e:      # a family that fails does not take the headline line with it
            out[name] = {"error": repr(e)}
        torch.cuda.empty_cache()
    S.reset_options()
    return out


_RESULT_FD = None


def _claim_stdout():
    """Keeps the process's standard output for the one JSON line: file descriptor 1 is pointed at standard error for
    everything else (RCCL prints a version banner to standard output when its first communicator is created)."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def _emit(line):
    sys.stdout.flush()
    if _RESULT_FD is None:
        print(line, flush=True)
    else:
        os.write(_RESULT_FD, (line + "\n").encode())


def family_main(a, rank, local_rank, world):
    """bench.py --family NAME [--gpus N]: the training step of one of the other model families as the benchmarked step.  One
    rank: the step replayed as one HIP graph.  N ranks (torch.distributed.run): every rank runs the step eagerly on its own
    batch (weak scaling), gradients averaged over the ranks by gan_amd.ddp.GradReducer, whose bucket all-reduces (RCCL)
    start from autograd hooks during the last backward pass of each phase and are waited for before the optimizer step."""
    import torch.distributed as dist
    import gan_amd
    from gan_amd import _lib
    from gan_amd.ddp import GradReducer
    from gan_amd.harness import steps as S
    device = f"cuda:{local_rank}"
    distributed = world > 1 or a.force_reducer
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    step = S.FAMILIES[a.family](device, reducer_factory=GradReducer if distributed else None, data_seed=6000 + rank)
    if distributed:
        for t in step.parameters() + step.buffers():
            gan_amd.ddp.broadcast(t.data, 0)
        if hasattr(step, "averaged_G"):
            step.averaged_G.load_state_dict(step.net_G.state_dict())
    use_graph = not distributed and not a.no_graph
    run = gan_amd.GraphedStep(step) if use_graph else step

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    if distributed:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    losses = step.losses.detach().float().cpu().tolist()
    if rank == 0:
        _lib.profile_enable(True)
    step()
    torch.cuda.synchronize()
    if rank == 0:
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        per = dt / a.steps
        dom = _dominant(prof, per)
        roofline = None
        if dom:
            roofline = {"bound": "mfma", "kernel": dom["dominant_kernel"], "achieved": dom["dominant_kernel_tflops"], "peak": PEAK_BF16_TFLOPS,
                        "unit": "TFLOP/s", "frac": dom["frac"], "traffic": None, "launches": dom["dominant_kernel_launches"],
                        "avg_launch_us": dom["dominant_kernel_avg_us"], "step_tflops": dom["step_tflops"],
                        "hbm_gb_s": _family_traffic(a.family, per),
                        "all_kernels": {k: {"launches": v["launches"], "avg_us": 1e3 * v["total_ms"] / max(v["launches"], 1),
                                            "tflops": v["flops"] / (v["total_ms"] * 1e-3) / 1e12 if v["total_ms"] > 0 else 0.0}
                                        for k, v in prof.items()}}
        _emit(json.dumps({
            "metric": "G+D train img/s", "value": world * step.batch * a.steps / dt, "unit": "img/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * per, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": step.what + "; activations stored in bf16 between layers, conv operands bf16 on MFMA with fp32 accumulate",
                       "family": a.family, "global_batch": world * step.batch, "parallelism": f"dp{world}", "hip_graph": bool(use_graph),
                       "gradient_all_reduce": "bucketed, started from autograd hooks during the last backward pass of a phase" if distributed else None,
                       "losses": losses, "losses_finite": all(x == x and abs(x) != float("inf") for x in losses)},
            "roofline": roofline}))
    if distributed:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="run the cycle eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true", help="skip the per-family figures (DCGAN, Pix2Pix, StyleGAN, BigGAN training steps)")
    ap.add_argument("--family", default=None, choices=["dcgan64_b128", "pix2pix256_b8", "pix2pix256_b64", "stylegan256_b32", "biggan128_b256"],
                    help="benchmark the training step of this model family instead of the WGAN-GP cycle (with --gpus N: data parallel, "
                         "that batch per GPU)")
    ap.add_argument("--overlap", action="store_true",
                    help="launch weight-gradient kernels on a second stream beside the data gradient (measured: about 1 %% faster)")
    ap.add_argument("--force-reducer", action="store_true", help="use the data-parallel gradient reducer even with one rank (test aid)")
    ap.add_argument("--ddp-eager", action="store_true",
                    help="data parallel: launch every kernel eagerly (default: forward/backward and optimizer steps are HIP "
                         "graphs, the gradient buckets are all-reduced between two replays)")
    ap.add_argument("--one-stream", action="store_true",
                    help="run the independent chains of a critic iteration (real + fake pass, gradient-penalty pass, the generator's "
                         "forward passes) one after the other on one stream instead of on streams of their own: the same launches, "
                         "every kernel alone on the GPU (for profiles; measured: about 20 %% slower)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="data parallel: BatchNorm statistics over all ranks (the single-device result on the global batch) "
                         "instead of per rank (what the reference's DataParallel does)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32x3"])
    ap.add_argument("--fp32-activations", action="store_true",
                    help="store the activations between the layers in fp32 (default: bf16 storage, fp32 accumulation and statistics)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    torch.cuda.set_device(local_rank)
    _claim_stdout()
    if a.family is not None:
        return family_main(a, rank, local_rank, world)
    import torch.distributed as dist
    distributed = world > 1 or a.force_reducer
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import gan_amd
    from gan_amd import _lib
    from gan_amd.ddp import GradReducer
    from gan_amd.harness import basic_gans
    from gan_amd.harness.basic_gans import WganGpTrainer, synth_inputs
    basic_gans.serialize_chains = bool(a.one_stream)
    gan_amd.set_precision(a.precision)
    gan_amd.set_backward_overlap(a.overlap)   # weight-gradient kernels on a second stream, beside the data gradient

    use_graph = not a.no_graph and not (distributed and a.ddp_eager)
    tr = WganGpTrainer(BATCH, HIDDEN, NZ, RES, device=f"cuda:{local_rank}", reducer_factory=(GradReducer if distributed else None),
                       capturable=use_graph, batch_real_fake=True, fused_adam=True,
                       activation_dtype=torch.float32 if a.fp32_activations else torch.bfloat16, two_streams=True)
    if a.sync_bn:
        if use_graph:
            raise SystemExit("--sync-bn exchanges statistics inside the forward pass: run it with --ddp-eager (collectives are never captured)")
        gan_amd.nn.convert_sync_batchnorm(tr.net_G)
    if distributed:  # replicas start from identical parameters
        for p in list(tr.net_G.parameters()) + list(tr.net_D.parameters()) + list(tr.net_G.buffers()) + list(tr.net_D.buffers()):
            gan_amd.ddp.broadcast(p.data, 0)
    gen = torch.Generator().manual_seed(5001 + rank)
    real, zs = synth_inputs(gen, BATCH, NZ, RES, REPS + 1)
    eps = torch.rand(REPS, BATCH, 1, 1, 1, generator=gen)
    tr.set_inputs(real, zs, eps)
    if use_graph:
        tr.capture(warmup=2)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.cycle()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        tr.cycle()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    if distributed:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    losses = tr.last.detach().float().cpu().tolist() if tr.graph is None else tr.static_out.detach().float().cpu().tolist()
    finite = all(x == x and abs(x) != float("inf") for x in losses)

    # ---- per-kernel timing of the same cycle (HIP events on the launch stream), eager, outside the timed region
    # One stream here: a kernel's duration is that of the kernel alone on the GPU, not of a kernel sharing it with another
    # chain.  The launches are those of the timed region (its chains serialised), e.g. one weight-gradient launch per layer.
    roofline = None
    basic_gans.serialize_chains = True
    if rank == 0:
        _lib.profile_enable(True)
    for _ in range(2):        # every rank runs the cycles: with more than one rank they contain collectives
        tr.cycle_eager()
    torch.cuda.synchronize()
    if rank == 0:
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        conv = {k: v for k, v in prof.items() if v["flops"] > 0}
        if conv:
            name = max(conv, key=lambda k: conv[k]["total_ms"])
            v = conv[name]
            achieved = v["flops"] / (v["total_ms"] * 1e-3) / 1e12
            traffic = None   # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_traffic.py), if recorded
            tpath = os.path.join(ROOT, "bench_traffic.json")   # copy of the newest profiles/r*_traffic.json
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
            roofline = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "launches": v["launches"],
                        "avg_launch_us": 1e3 * v["total_ms"] / v["launches"], "avg_launch_gflop": v["flops"] / v["launches"] / 1e9,
                        "all_kernels": {k: {"launches": q["launches"], "avg_us": 1e3 * q["total_ms"] / max(q["launches"], 1),
                                            "tflops": (q["flops"] / (q["total_ms"] * 1e-3) / 1e12) if q["total_ms"] > 0 else 0.0}
                                        for k, q in prof.items()}}

    if rank == 0:
        out = {
            "metric": "G+D train img/s", "value": world * BATCH * a.steps / dt, "unit": "img/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if a.precision == "bf16" else "bf16x3", "data": "synthetic",
            "config": {"workload": "BasicGANs WGAN-GP 64x64 (hidden 64, nz 100), batch 256 per GPU; step = 5 critic iterations "
                                   "(real and fake critic passes batched into one 512-sample pass; gradient-penalty double backward) + 1 generator iteration, Adam; activations " +
                                   ("fp32" if a.fp32_activations else "stored in bf16 between layers (images, scores, normalisation statistics and all accumulation fp32)") +
                                   ", NHWC; conv operands bf16 on MFMA with fp32 accumulate",
                       "activation_storage": "fp32" if a.fp32_activations else "bf16",
                       "global_batch": world * BATCH, "parallelism": f"dp{world}", "hip_graph": bool(use_graph), "concurrent_chains": not a.one_stream, "batchnorm_statistics": "all ranks" if a.sync_bn else "per rank",
                       "losses_finite": finite},
            "roofline": roofline,
        }
        if world == 1 and not a.no_families:
            del tr
            torch.cuda.empty_cache()
            out["families"] = families(f"cuda:{local_rank}")
        if world == 1 and not a.no_cpu_baseline:
            threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
            try:
                out["cpu_baseline"] = cpu_baseline(threads)
            except Exception as e:  # the GPU number stands on its own
                out["cpu_baseline"] = {"error": repr(e)}
        _emit(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
