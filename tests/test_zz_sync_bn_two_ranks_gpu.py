"""BatchNorm with statistics over the ranks of a process group, through the HIP kernels: two processes on one GPU (gloo
carries the small statistic vectors), unequal shards of one batch; outputs, input gradients, parameter gradients and
running statistics must be those of torch's BatchNorm2d + activation on the whole batch (float64 on the CPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

SHAPE = (8, 24, 12, 10)
SPLIT = (3, 5)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _data(dtype):
    g = torch.Generator().manual_seed(23)
    x = torch.randn(*SHAPE, generator=g) * 1.7 + torch.linspace(-40, 40, SHAPE[1]).view(1, -1, 1, 1)   # |mean| >> std in the outer channels
    gy = torch.randn(*SHAPE, generator=g)
    w = torch.rand(SHAPE[1], generator=g) + 0.5
    b = torch.randn(SHAPE[1], generator=g) * 0.2
    if dtype == torch.bfloat16:
        x, gy = x.bfloat16().float(), gy.bfloat16().float()
    return x, gy, w, b


def _reference(dtype, act):
    x, gy, w, b = _data(dtype)
    bn = torch.nn.BatchNorm2d(SHAPE[1]).double()
    bn.weight.data.copy_(w); bn.bias.data.copy_(b)
    xd = x.double().requires_grad_(True)
    y = bn(xd)
    y = torch.relu(y) if act == "relu" else (torch.nn.functional.leaky_relu(y, 0.2) if act == "leaky_relu" else y)
    y.backward(gy.double())
    return y.detach(), xd.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var


def _worker(rank, world, port, q, dtype_name, act):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gan_amd
        dtype = getattr(torch, dtype_name)
        x, gy, w, b = _data(dtype)
        lo = sum(SPLIT[:rank]); sl = slice(lo, lo + SPLIT[rank])
        bn = gan_amd.nn.BatchNormAct2d(SHAPE[1], activation=act, sync_stats=True).cuda()
        bn.weight.data.copy_(w); bn.bias.data.copy_(b)
        xs = x[sl].cuda().to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = bn(xs)
        y.backward(gy[sl].cuda().to(dtype))
        torch.cuda.synchronize()
        q.put((rank, y.detach().float().cpu().numpy(), xs.grad.float().cpu().numpy(), bn.weight.grad.cpu().numpy(), bn.bias.grad.cpu().numpy(),
               bn.running_mean.cpu().numpy(), bn.running_var.cpu().numpy(), int(bn.num_batches_tracked)))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name,tol", [("float32", 2e-5), ("bfloat16", 8e-3)])
@pytest.mark.parametrize("act", [None, "leaky_relu"])
def test_statistics_over_two_ranks_match_the_whole_batch(dtype_name, tol, act):
    y, gx, gw, gb, rm, rv = _reference(getattr(torch, dtype_name), act)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, dtype_name, act)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=150) for _ in range(2)], key=lambda r: r[0])
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for p in procs:            # a rank that hangs must not take the rest of the suite with it
            if p.is_alive():
                p.kill()

    def rel(a, b):
        a = torch.as_tensor(a).double()
        return ((a - b).norm() / b.norm()).item()

    sum_gw = sum(torch.from_numpy(r[3]).double() for r in res)
    sum_gb = sum(torch.from_numpy(r[4]).double() for r in res)
    assert rel(sum_gw, gw) < tol and rel(sum_gb, gb) < tol
    for rank, ys, gxs, _, _, rms, rvs, nbt in res:
        lo = sum(SPLIT[:rank]); sl = slice(lo, lo + SPLIT[rank])
        assert rel(ys, y[sl]) < tol, rank
        assert rel(gxs, gx[sl]) < 2 * tol, rank
        assert rel(rms, rm) < 1e-5 and rel(rvs, rv) < 1e-4 and nbt == 1       # every rank holds the whole batch's running statistics


@pytest.mark.gpu
def test_one_rank_is_plain_batchnorm():
    import gan_amd
    x, gy, w, b = _data(torch.float32)
    outs = []
    for sync in (False, True):
        bn = gan_amd.nn.BatchNormAct2d(SHAPE[1], activation="relu", sync_stats=sync).cuda()
        bn.weight.data.copy_(w); bn.bias.data.copy_(b)
        xs = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = bn(xs)
        y.backward(gy.cuda())
        outs.append((y.detach(), xs.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var))
    for a, c in zip(*outs):
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-5)
