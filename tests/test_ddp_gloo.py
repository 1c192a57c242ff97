"""Data-parallel gradient reducer on CPU with the gloo backend, world size 2: the averaged gradients of two
half-batches equal the single-process gradients of the full batch (the WGAN critic has no batch coupling:
InstanceNorm is per sample, and the losses are batch means over equal shards; the generator's BatchNorm is put
in eval mode because batch statistics over a shard differ from those over the full batch by construction)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_models as R


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _critic_grads(net_G, net_D, real, z, eps, reducer=None):
    from gan_amd.harness.basic_gans import wgan_critic_phase
    if reducer is not None:
        reducer.zero_grad()
    else:
        net_D.zero_grad()
    wgan_critic_phase(net_G, net_D, real, z, eps, 10.0, reducer.arm if reducer is not None else None)
    if reducer is not None:
        reducer.finish()
    return [p.grad.detach().clone() for p in net_D.parameters()]


def _make(seed=7):
    # float64 so that the comparison is not limited by the conditioning of the penalty's double backward
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(seed)
    G = R.Generator(16, 4, 3, resolution=16); D = R.Discriminator(3, 4, resolution=16, is_wgan=True)
    G.apply(R.weights_init); D.apply(R.weights_init)
    G.eval()   # the generator's BatchNorm couples the samples of a batch; running statistics remove that coupling
    g = torch.Generator().manual_seed(11)
    real = torch.rand(4, 3, 16, 16, generator=g) * 2 - 1
    z = torch.randn(4, 16, 1, 1, generator=g)
    eps = torch.rand(4, 1, 1, 1, generator=g)
    torch.set_default_dtype(torch.float32)
    return G, D, real, z, eps


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gan_amd.ddp import GradReducer
        torch.set_num_threads(1)
        G, D, real, z, eps = _make()
        red = GradReducer(D, bucket_bytes=1024)   # small buckets: several collectives, some triggered by hooks
        sl = slice(rank * 2, rank * 2 + 2)
        grads = _critic_grads(G, D, real[sl], z[sl], eps[sl], red)
        # second phase re-uses the buckets: must give the same result again
        grads2 = _critic_grads(G, D, real[sl], z[sl], eps[sl], red)
        ok = all(torch.allclose(a, b, rtol=1e-10, atol=1e-14) for a, b in zip(grads, grads2))
        # plain form (forward/backward replayed as a graph, buckets reduced between replays): same numbers
        red.zero_grad()
        from gan_amd.harness.basic_gans import wgan_critic_phase
        wgan_critic_phase(G, D, real[sl], z[sl], eps[sl], 10.0, None, True)   # real + fake as one batch, no hooks armed
        red.all_reduce(); red.average_()
        grads3 = [p.grad.detach().clone() for p in D.parameters()]
        ok = ok and all(torch.allclose(a, b, rtol=1e-8, atol=1e-12) for a, b in zip(grads, grads3))
        # a phase that never arms the hooks (gradients added by the caller), twice in a row: finish() must reduce both times
        for _ in range(2):
            red.zero_grad()
            wgan_critic_phase(G, D, real[sl], z[sl], eps[sl], 10.0, None, True)
            red.finish()
            grads4 = [p.grad.detach().clone() for p in D.parameters()]
            ok = ok and all(torch.allclose(a, b, rtol=1e-8, atol=1e-12) for a, b in zip(grads, grads4))
        q.put((rank, [g.numpy() for g in grads], len(red.buckets), ok))
    finally:
        dist.destroy_process_group()


def test_grad_reducer_world2_matches_full_batch():
    G, D, real, z, eps = _make()
    ref = _critic_grads(G, D, real, z, eps)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, grads, nb, ok in res:
        assert nb >= 2 and ok
        for g, r in zip(grads, ref):
            g = torch.from_numpy(g)
            assert (g - r).norm() <= 1e-9 * r.norm() + 1e-14, rank


def test_grad_reducer_single_process_is_identity():
    from gan_amd.ddp import GradReducer
    G, D, real, z, eps = _make()
    ref = _critic_grads(G, D, real, z, eps)
    G2, D2, *_ = _make()
    red = GradReducer(D2)
    got = _critic_grads(G2, D2, real, z, eps, red)
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-14)
    # zero_grad(set_to_none) from user code must not break the flat views
    for p in D2.parameters():
        p.grad = None
    got = _critic_grads(G2, D2, real, z, eps, red)
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, rtol=1e-10, atol=1e-14)
