"""Developer tool: a convolution with a channel count that is not a multiple of 8 (StyleGAN's 513 -> 512 layer) in each
storage type: forward, input gradient and weight gradient against torch on the same (rounded) values, on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gan_amd as gan
import gan_amd.functional as GF

torch.manual_seed(1)
for cin, cout, k, p, hw in ((513, 512, 3, 1, 4), (512, 512, 4, 0, 4), (13, 24, 3, 1, 8)):
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(4, cin, hw, hw, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(cout, cin, k, k, device="cuda") * 0.05).requires_grad_(True)
        b = torch.zeros(cout, device="cuda", requires_grad=True)
        y = GF.conv2d(x, w, b, padding=p, activation="leaky_relu")
        gy = torch.randn_like(y)
        gx, gw, gb = torch.autograd.grad(y, (x, w, b), gy)
        xr = x.detach().float().requires_grad_(True)
        wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
        yr = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xr.to(torch.bfloat16).float(), wr, None, padding=p), 0.2)
        gxr, gwr = torch.autograd.grad(yr, (xr, wr), gy.float())
        rel = lambda a, b_: ((a.float() - b_).norm() / b_.norm()).item()
        print(f"cin {cin} cout {cout} k{k} {dt}: y {rel(y, yr):.2e} nan={torch.isnan(y.float()).any().item()}  gx {rel(gx, gxr):.2e} nan={torch.isnan(gx.float()).any().item()}"
              f"  gw {rel(gw, gwr):.2e} nan={torch.isnan(gw).any().item()}  gb nan={torch.isnan(gb).any().item()}", flush=True)

# second order (gradient-penalty shape): d/dw, d/dx of |d sum(y) / dx|^2 through two stacked layers
for cin, cmid, hw in ((513, 512, 4), (13, 24, 8)):
    for dt in (torch.float32, torch.bfloat16):
        x0 = torch.randn(4, cin, hw, hw, device="cuda")
        w1 = (torch.randn(cmid, cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
        w2 = (torch.randn(cmid, cmid, 3, 3, device="cuda") * 0.05).requires_grad_(True)

        def penalty(conv, x, cast):
            h = conv(x, w1, 1)
            y = conv(h, w2, 1)
            g, = torch.autograd.grad(y.float().sum(), x, create_graph=True)
            return (g.float() ** 2).sum()

        x = x0.to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        pen = penalty(lambda t, w, p: GF.conv2d(t, w, None, padding=p, activation="leaky_relu"), x, dt)
        g1, g2 = torch.autograd.grad(pen, (w1, w2))
        xr = x0.to(dt).float().requires_grad_(True)
        penr = penalty(lambda t, w, p: torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(t, w, None, padding=p), 0.2), xr, dt)
        g1r, g2r = torch.autograd.grad(penr, (w1, w2))
        print(f"second order cin {cin} {dt}: penalty {pen.item():.5g} vs {penr.item():.5g}  gw1 {rel(g1, g1r):.2e} nan={torch.isnan(g1).any().item()}"
              f"  gw2 {rel(g2, g2r):.2e} nan={torch.isnan(g2).any().item()}", flush=True)
