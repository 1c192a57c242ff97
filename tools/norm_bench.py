"""Developer tool: InstanceNorm kernels (forward, backward, backward of the backward) at the benchmark's critic shapes:
time per launch (HIP events around the launch) and bytes moved per second (each tensor counted once)."""
import sys
import torch
sys.path.insert(0, ".")
from gan_amd import ops, _lib

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    prof = _lib.profile_collect()
    _lib.profile_enable(False)
    return sum(v["total_ms"] for v in prof.values()) / n * 1e3

for N in (256, 512):
    for C, H in ((128, 16), (256, 8), (512, 4)):
        x = torch.randn(N, C, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        g = torch.randn_like(x); u = torch.randn_like(x)
        mb = x.numel() * 2 / 1e6
        t0 = timeit(lambda: ops.instnorm_act(x, 1e-5, _lib.GC_ACT_LRELU, 0.2))
        t1 = timeit(lambda: ops.instnorm_act_bwd(x, g, 1e-5, _lib.GC_ACT_LRELU, 0.2))
        t2 = timeit(lambda: ops.instnorm_act_bwd_bwd(x, g, u, 1e-5, _lib.GC_ACT_LRELU, 0.2))
        print(f"N {N:4d} C {C:4d} {H:2d}x{H:<2d} {mb:6.1f} MB/tensor | fwd {t0:6.1f} us {2*mb/t0:6.2f} TB/s | bwd {t1:6.1f} us {3*mb/t1:6.2f} TB/s | bwd_bwd {t2:6.1f} us {5*mb/t2:6.2f} TB/s")
