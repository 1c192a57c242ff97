"""Developer tool: the attention kernels at the BigGAN shapes (batch 256, 4096 positions) against torch's fused
attention call on the same bf16 operands (zero-padded to one width, as that call needs).  GPU box:
    python tools/attention_bench.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import gan_amd as gan

B, N = 256, 4096
for d, C in ((8, 64), (16, 128)):
    k, q = (torch.randn(B, N, d, device="cuda").to(torch.bfloat16).requires_grad_(True) for _ in range(2))
    v = torch.randn(B, N, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)

    def ours():
        return gan.ops.attention(k, q, v)[0]

    def sdpa():
        return F.scaled_dot_product_attention(F.pad(k, (0, C - d))[:, None], F.pad(q, (0, C - d))[:, None], v[:, None], scale=1.0)[:, 0]

    for name, fn in (("gan_amd.ops.attention", ours), ("torch fused attention", sdpa)):
        def timed(f, n=5):
            for _ in range(2):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        tf = timed(fn)
        o = fn()
        tb = timed(lambda: torch.autograd.grad(o, (k, q, v), go, retain_graph=True))
        fl = 2.0 * B * N * N * (d + C)
        print(f"d={d:2d} C={C:3d} {name:24s} forward {tf:7.2f} ms ({fl / tf / 1e9:6.0f} TFLOP/s)   backward {tb:7.2f} ms ({2.5 * fl / tb / 1e9:6.0f} TFLOP/s)", flush=True)
