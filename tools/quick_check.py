"""Developer check: the three conv ops against stock torch CPU convolutions on a list of geometries."""
import sys, time
import torch
import torch.nn.functional as TF
sys.path.insert(0, ".")
import gan_amd
from gan_amd import ops, _lib

def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

CASES = [  # N, Cin, H, W, Cout, k, s, p
    (2, 3, 16, 16, 16, 4, 2, 1), (2, 32, 16, 16, 64, 4, 2, 1), (3, 64, 4, 4, 1, 4, 1, 0), (2, 16, 9, 9, 24, 4, 1, 1),
    (2, 24, 10, 12, 40, 3, 1, 1), (2, 40, 7, 5, 3, 1, 1, 0), (1, 513, 4, 4, 8, 3, 1, 1), (4, 64, 32, 32, 128, 4, 2, 1),
    (4, 128, 16, 16, 256, 4, 2, 1), (8, 256, 8, 8, 512, 4, 2, 1), (3, 100, 4, 4, 32, 4, 1, 0), (2, 8, 10, 10, 8, 3, 2, 1),
    (2, 128, 31, 31, 96, 4, 1, 1), (2, 6, 32, 32, 64, 4, 2, 1),
]
torch.manual_seed(0)
bad = 0
for prec_name in ("fp32x3", "bf16"):
    prec = {"bf16": _lib.GC_PREC_BF16, "fp32x3": _lib.GC_PREC_F32X3}[prec_name]
    tol = 2e-5 if prec_name == "fp32x3" else 6e-3
    for (N, Cin, H, W, Cout, k, s, p) in CASES:
        x = torch.randn(N, Cin, H, W); w = torch.randn(Cout, Cin, k, k) * 0.1; b = torch.randn(Cout)
        y = TF.conv2d(x, w, b, s, p); gy = torch.randn_like(y)
        gx = torch.nn.grad.conv2d_input(x.shape, w, gy, s, p)
        gw = torch.nn.grad.conv2d_weight(x, w.shape, gy, s, p)
        xd, wd, bd, gyd = x.cuda(), w.cuda(), b.cuda(), gy.cuda()
        y2 = ops.conv_fprop(xd, wd, bd, s, p, prec, 0)
        gx2 = ops.conv_dgrad(gyd, wd, None, H, W, s, p, prec, 0)
        gw2 = ops.conv_wgrad(xd, gyd, k, k, s, p, prec)
        torch.cuda.synchronize()
        e = (rel(y2, y), rel(gx2, gx), rel(gw2, gw))
        flag = "" if max(e) < tol else "  <-- FAIL"
        bad += max(e) >= tol
        print(f"{prec_name:7s} N{N} C{Cin}->{Cout} {H}x{W} k{k}s{s}p{p}: fprop {e[0]:.2e} dgrad {e[1]:.2e} wgrad {e[2]:.2e}{flag}")
# bf16 storage (LDS-DMA fast path where channels allow): compare against CPU conv on the same bf16-rounded inputs
for (N, Cin, H, W, Cout, k, s, p) in CASES:
    prec = _lib.GC_PREC_BF16
    x = torch.randn(N, Cin, H, W).bfloat16(); w = (torch.randn(Cout, Cin, k, k) * 0.1).bfloat16().float(); b = torch.randn(Cout)
    y = TF.conv2d(x.float(), w, b, s, p); gy = torch.randn_like(y).bfloat16()
    gx = torch.nn.grad.conv2d_input(x.shape, w, gy.float(), s, p)
    gw = torch.nn.grad.conv2d_weight(x.float(), w.shape, gy.float(), s, p)
    y2 = ops.conv_fprop(x.cuda(), w.cuda(), b.cuda(), s, p, prec, 0)
    gx2 = ops.conv_dgrad(gy.cuda(), w.cuda(), None, H, W, s, p, prec, 0)
    gw2 = ops.conv_wgrad(x.cuda(), gy.cuda(), k, k, s, p, prec)
    torch.cuda.synchronize()
    e = (rel(y2.float(), y), rel(gx2.float(), gx), rel(gw2, gw))
    flag = "" if max(e) < 4e-3 else "  <-- FAIL"
    bad += max(e) >= 4e-3
    print(f"bf16io  N{N} C{Cin}->{Cout} {H}x{W} k{k}s{s}p{p}: fprop {e[0]:.2e} dgrad {e[1]:.2e} wgrad {e[2]:.2e}{flag}")
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
