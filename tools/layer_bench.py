"""Developer tool: per-layer timing of the three conv passes at the benchmark's shapes (HIP events, median)."""
import sys, argparse
import torch
sys.path.insert(0, ".")
import gan_amd
from gan_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--prec", default="bf16")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--set", default="wgan")
a = ap.parse_args()
prec = {"bf16": _lib.GC_PREC_BF16, "fp32x3": _lib.GC_PREC_F32X3}[a.prec]
dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
SETS = {
    "wgan": [(3, 64, 64, 4, 2, 1), (64, 32, 128, 4, 2, 1), (128, 16, 256, 4, 2, 1), (256, 8, 512, 4, 2, 1), (512, 4, 1, 4, 1, 0), (512, 4, 100, 4, 1, 0)],
    "ksweep": [(64, 32, 128, 4, 2, 1), (128, 32, 128, 4, 2, 1), (256, 32, 128, 4, 2, 1), (512, 32, 128, 4, 2, 1), (1024, 32, 128, 4, 2, 1),
               (64, 32, 128, 3, 1, 1), (64, 32, 128, 1, 1, 0), (128, 32, 128, 1, 1, 0), (512, 32, 128, 1, 1, 0)],
    "big": [(256, 32, 512, 4, 2, 1), (512, 64, 512, 3, 1, 1), (128, 128, 128, 3, 1, 1), (512, 16, 512, 4, 2, 1),
            (512, 32, 512, 3, 1, 1), (256, 64, 256, 3, 1, 1), (1024, 16, 1024, 3, 1, 1)],
    # the StyleGAN-256 layers that carry its FLOPs (run with --batch 32 --dtype bf16)
    "sg": [(64, 256, 128, 3, 1, 1), (128, 128, 256, 3, 1, 1), (256, 64, 512, 3, 1, 1), (512, 64, 512, 3, 1, 1), (512, 32, 512, 3, 1, 1),
           (512, 64, 256, 3, 1, 1), (256, 64, 256, 3, 1, 1), (128, 128, 128, 3, 1, 1), (64, 256, 64, 3, 1, 1), (128, 256, 128, 4, 2, 1),
           (256, 128, 256, 4, 2, 1)],
}
B = a.batch

def timeit(fn, n=20):
    """Median over n calls of the library kernels' own time (events around each launch), excluding weight packing."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    tot = sum(v["total_ms"] for k, v in prof.items() if "pack_weight" not in k)
    timeit.last = {k.replace("gc::", "").replace("unsigned short", "u16"): v["total_ms"] / n * 1e3 for k, v in prof.items() if "pack_weight" not in k}
    return tot / n * 1e3

def timeit_wall(fn, n=20):
    for _ in range(3): fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in evs:
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return ts[len(ts) // 2] * 1e3  # us

print(f"batch {B} prec {a.prec} dtype {a.dtype}")
for (Cin, H, Cout, k, s, p) in SETS[a.set]:
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn(B, Cin, H, H, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    gy = torch.randn(B, Cout, Ho, Ho, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    flop = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    key = ops.new_weight_key()
    wp = torch.nn.Parameter(w)
    t_f = timeit(lambda: ops.conv_fprop(x, wp, None, s, p, prec, key))
    t_d = timeit(lambda: ops.conv_dgrad(gy, wp, None, H, H, s, p, prec, key))
    kf, kd = dict(timeit.last), None
    t_w = timeit(lambda: ops.conv_wgrad(x, gy, k, k, s, p, prec))
    kw = dict(timeit.last)
    print(f"C{Cin:4d}->{Cout:4d} {H:3d}x{H:<3d} k{k}s{s}p{p} {flop/1e9:8.2f} GFLOP | fprop {t_f:8.1f} us {flop/t_f/1e6:7.1f} TF | dgrad {t_d:8.1f} us {flop/t_d/1e6:7.1f} TF | wgrad {t_w:8.1f} us {flop/t_w/1e6:7.1f} TF | " +
          " + ".join(f"{k} {v:.0f}" for k, v in kw.items()))
