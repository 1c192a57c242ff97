"""Probe: does torch's fused attention run the BigGAN self-attention shape (4096 positions, key/query width C/8, value
width C) on this GPU, and how long does forward + backward take against bmm + softmax?"""
import time
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

torch.manual_seed(0)
b, N = 64, 4096
for d, C in ((8, 64), (16, 128)):
    for dt in (torch.float32, torch.bfloat16):
        key = torch.randn(b, N, d, device="cuda", dtype=dt, requires_grad=True)
        query = torch.randn(b, N, d, device="cuda", dtype=dt, requires_grad=True)
        value = torch.randn(b, N, C, device="cuda", dtype=dt, requires_grad=True)

        def plain():
            a = torch.softmax(torch.bmm(key, query.transpose(1, 2)), dim=-1)
            return torch.bmm(a, value)

        def fused(pad):
            q, k = key, query
            if pad:
                q, k = F.pad(q, (0, pad - d)), F.pad(k, (0, pad - d))
            return F.scaled_dot_product_attention(q[:, None], k[:, None], value[:, None], scale=1.0)[:, 0]

        def run(fn, tag):
            try:
                for _ in range(2):
                    o = fn(); o.sum().backward()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    o = fn(); o.sum().backward()
                torch.cuda.synchronize()
                print(f"d={d} C={C} {dt} {tag:28s} {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
                return o.detach()
            except Exception as e:
                print(f"d={d} C={C} {dt} {tag:28s} failed: {str(e)[:150]}", flush=True)

        ref = run(plain, "bmm+softmax")
        for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
            for pad in (0, C):
                with sdpa_kernel(be):
                    o = run(lambda: fused(pad), f"{be.name[:9]} pad={pad}")
                if o is not None and ref is not None:
                    print("      max abs diff", (o.float() - ref.float()).abs().max().item())
