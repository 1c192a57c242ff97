"""Developer tool: the memory-bound first-layer kernels of the critic (3 -> 64 channels, 64 x 64 images) at the
benchmark's batch sizes: time per launch and bytes moved per second.   python tools/thin_bench.py [--batch 256]"""
import argparse
import sys
sys.path.insert(0, ".")
import torch
import gan_amd
from gan_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, nargs="*", default=[256, 512])
a = ap.parse_args()
P = _lib.GC_PREC_BF16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    rows = {k: v["total_ms"] / n * 1e3 for k, v in prof.items() if "pack_weight" not in k}
    return sum(rows.values()), rows


for B in a.batch:
    x = torch.randn(B, 3, 64, 64, device="cuda").contiguous(memory_format=torch.channels_last)
    xb = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(64, 3, 4, 4, device="cuda") * 0.05)
    gy = torch.randn(B, 64, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    key = ops.new_weight_key()
    mb_out = B * 64 * 32 * 32 * 2 / 1e6
    cases = [
        ("fprop  f32 image -> bf16, LeakyReLU", lambda: ops.conv_fprop(x, w, None, 2, 1, P, key, 2, 0.2, 1), B * 3 * 64 * 64 * 4 / 1e6 + mb_out),
        ("fprop  bf16 image -> bf16", lambda: ops.conv_fprop(xb, w, None, 2, 1, P, key, 2, 0.2, 1), B * 3 * 64 * 64 * 2 / 1e6 + mb_out),
        ("dgrad  bf16 -> f32 image", lambda: ops.conv_dgrad(gy, w, None, 64, 64, 2, 1, P, key, 0, 0.0, 0), B * 3 * 64 * 64 * 4 / 1e6 + mb_out),
        ("wgrad  f32 image, bf16 gradient", lambda: ops.conv_wgrad(x, gy, 4, 4, 2, 1, P), B * 3 * 64 * 64 * 4 / 1e6 + mb_out),
        ("wgrad  bf16 image, bf16 gradient", lambda: ops.conv_wgrad(xb, gy, 4, 4, 2, 1, P), B * 3 * 64 * 64 * 2 / 1e6 + mb_out),
    ]
    for name, fn, mb in cases:
        try:
            t, rows = timeit(fn)
        except Exception as e:       # a dtype pair the op does not take
            print(f"B {B:4d} {name:36s} -- {type(e).__name__}: {str(e)[:80]}")
            continue
        print(f"B {B:4d} {name:36s} {t:7.1f} us  {mb / t:6.2f} TB/s   " + ", ".join(f"{k.replace('gc::', '')} {v:.1f}" for k, v in rows.items()))
