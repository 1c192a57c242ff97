"""Developer probe: a few launches of the attention kernels at the BigGAN shape, for counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gan_amd as gan
B, N = 256, 4096
for d, C in ((8, 64), (16, 128)):
    k, q = (torch.randn(B, N, d, device="cuda").to(torch.bfloat16).requires_grad_(True) for _ in range(2))
    v = torch.randn(B, N, C, device="cuda").to(torch.bfloat16).requires_grad_(True)
    go = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        o = gan.ops.attention(k, q, v)[0]
        torch.autograd.grad(o, (k, q, v), go)
torch.cuda.synchronize()
