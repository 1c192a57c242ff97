"""Times the 1 x 1 from-RGB convolution (3 -> 64 channels, 256 x 256, batch 32) per output type."""
import sys
sys.path.insert(0, ".")
import torch
import gan_amd
from gan_amd import _lib

x = torch.randn(32, 3, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
w = torch.randn(64, 3, 1, 1, device="cuda")
b = torch.randn(64, device="cuda")
for od in (torch.bfloat16, torch.float32):
    for xin in (x, x.to(torch.bfloat16)):
        fn = lambda: gan_amd.functional.conv2d(xin, w, b, activation="leaky_relu", out_dtype=od)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        print(xin.dtype, "->", od, {k: round(v["total_ms"] / v["launches"] * 1e3, 1) for k, v in prof.items()})
