"""Probe: torch's multi-block reduction recorded in a HIP graph, replayed between eager launches of the same reduction."""
import torch
x = torch.randn(2, 16, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
z = torch.randn(2, 16, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        y = x.sum((0, 2, 3), dtype=torch.float32)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = x.sum((0, 2, 3), dtype=torch.float32)
for i in range(6):
    x.copy_(torch.randn_like(x))
    want = x.double().sum((0, 2, 3))
    g.replay()
    got = y.clone()
    for _ in range(3):
        e = z.sum((0, 2, 3), dtype=torch.float32)      # eager launches of the same reduction in between
    torch.cuda.synchronize()
    print(i, "max abs error of the replayed sum", float((got.double() - want).abs().max()), flush=True)
