"""The data-parallel training steps of the Pix2Pix, StyleGAN and BigGAN harnesses (gan_amd/harness/steps.py) with two ranks on
one GPU, gloo carrying the gradient buckets: discriminator and generator each have their own GradReducer, whose buckets are
all-reduced from autograd hooks during the last backward pass of a phase (StyleGAN's critic phase accumulates over three
backward passes and reduces once).  Both ranks start from the same parameters and see different data; after every step
their parameters must be identical, finite and different from the start.  Reduced sizes (Pix2Pix ch 8, batch 2; StyleGAN
progressive step 2 = 16x16 at full width, batch 4; BigGAN ch 8, batch 4): the step logic is that of the full-size steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

FAMILIES = {
    "pix2pix": lambda S, **kw: S.pix2pix_unet("cuda:0", batch=2, ch=8, **kw),
    "stylegan": lambda S, **kw: S.stylegan_v1("cuda:0", batch=4, pg=2, **kw),
    "biggan": lambda S, **kw: S.biggan_128("cuda:0", batch=4, ch=8, n_classes=4, **kw),
}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _flat(net):
    return torch.cat([p.detach().float().flatten() for p in net.parameters()])


def _worker(rank, world, port, q, family):
    import faulthandler
    root = os.environ.get("GRAFT_REPO_ROOT", ".")
    log = open(os.path.join(root, "gpurun_out", f"ddp_family_{family}_rank{rank}.log"), "w") if os.path.isdir(os.path.join(root, "gpurun_out")) else None
    faulthandler.dump_traceback_later(150, exit=True, file=log if log is not None else None)   # a hung rank reports where it waits and ends
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gan_amd
        from gan_amd.ddp import GradReducer
        from gan_amd.harness import steps as S
        step = FAMILIES[family](S, reducer_factory=GradReducer, data_seed=900 + rank)
        torch.cuda.synchronize()
        for t in step.parameters() + step.buffers():
            gan_amd.ddp.broadcast(t.data, 0)
        torch.cuda.synchronize()
        d0, g0 = _flat(step.net_D).clone(), _flat(step.net_G).clone()
        sums = []
        for _ in range(2):
            step()
            torch.cuda.synchronize()
            d, g = _flat(step.net_D), _flat(step.net_G)
            sums.append((d.double().sum().item(), d.double().abs().sum().item(), g.double().sum().item(), g.double().abs().sum().item(),
                         bool(torch.isfinite(step.losses).all()) and bool(torch.isfinite(d).all()) and bool(torch.isfinite(g).all()),
                         float((d - d0).abs().max()), float((g - g0).abs().max()), step.losses.cpu().tolist()))
        q.put((rank, sums))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("family", list(FAMILIES))
def test_family_step_keeps_two_ranks_in_step(family):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, family)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = dict(q.get(timeout=200) for _ in range(2))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    for k in range(2):
        a, b = res[0][k], res[1][k]
        assert a[4] and b[4], (k, a, b)                         # finite losses and parameters on both ranks
        assert a[5] > 0 and a[6] > 0                            # both networks moved
        assert a[7] != b[7]                                     # the ranks saw different data
        assert a[0] == b[0] and a[1] == b[1], (k, a, b)         # identical discriminators
        assert a[2] == b[2] and a[3] == b[3], (k, a, b)         # identical generators
