"""Probe: torch.cuda.graph capture with a stream forked from a forked stream (the plain-HIP form of the same capture,
tools/probe/nested_fork_capture.hip, records and replays correctly).
usage: python tools/probe/nested_fork_torch.py prealloc|alloc|alloc_flat|wgrad
  prealloc    every tensor exists before the capture; the streams only launch kernels
  alloc       tensors are allocated inside the capture on the nested stream (torch's caching allocator, graph pool)
  alloc_flat  the same allocations with both streams forked from the capture's origin stream
  wgrad       the library's own case: a convolution backward with the weight-gradient side stream enabled, run on a
              stream that is itself forked inside the capture (the guard in gan_amd.ops._SideLaunch switched off)"""
import faulthandler
import os
import sys

import torch

faulthandler.enable()          # a crash inside the runtime prints the Python call it happened in

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1] if len(sys.argv) > 1 else "alloc"
dev = torch.device("cuda:0")
side, inner = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros(1 << 20, device=dev)
c = torch.zeros(1 << 20, device=dev)


def body():
    main = torch.cuda.current_stream(dev)
    a.add_(1)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        if mode == "prealloc":
            b.add_(1)
        else:
            t = torch.ones(1 << 20, device=dev)
            b.add_(t)
        (inner.wait_stream(side) if mode != "alloc_flat" else inner.wait_stream(main))
        with torch.cuda.stream(inner):
            if mode == "prealloc":
                c.add_(1)
            else:
                u = torch.ones(1 << 20, device=dev)
                c.add_(u)
        (side if mode != "alloc_flat" else main).wait_stream(inner)
        b.add_(1)
    main.wait_stream(side)
    a.add_(1)


def wgrad_body():
    import gan_amd
    from gan_amd import ops
    conv = wgrad_body.conv
    x = wgrad_body.x
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        y = conv(x)
        (y.float() ** 2).sum().backward()
    main.wait_stream(side)


if mode == "wgrad":
    import gan_amd
    from gan_amd import ops
    wgrad_body.conv = gan_amd.nn.Conv2d(64, 64, 3, 1, 1, bias=False).cuda()
    wgrad_body.x = torch.randn(8, 64, 16, 16, device=dev, requires_grad=True)
    gan_amd.set_backward_overlap(True)
    fn = wgrad_body
else:
    fn = body

s = torch.cuda.Stream(dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    fn()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("warm-up done", flush=True)
g = torch.cuda.CUDAGraph()
if mode == "wgrad":
    ops.clear_weight_cache(); ops.clear_cast_cache()
    # name the forked stream as the capture's origin: the library then forks its weight-gradient stream from it
    with torch.cuda.graph(g), ops.capturing_on(side):
        fn()
else:
    with torch.cuda.graph(g):
        fn()
print("capture ended", flush=True)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print(f"{mode}: replayed; a {a[0].item()} b {b[0].item()} c {c[0].item()}", flush=True)
