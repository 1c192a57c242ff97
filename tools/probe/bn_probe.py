import sys
sys.path.insert(0, ".")
import torch, gan_amd
from gan_amd import _lib, nn as gnn
for shape in [(8, 512, 4, 4), (8, 512, 8, 8), (8, 512, 16, 16)]:
    for dt in (torch.float32, torch.bfloat16):
        bn = gnn.BatchNormAct2d(shape[1], activation="relu").cuda()
        x = torch.randn(*shape, device="cuda").to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        _lib.profile_enable(True)
        y = bn(x)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile_enable(False)
        print(shape, dt, {k: (v["launches"], round(v["total_ms"] * 1e3, 1)) for k, v in prof.items()})
