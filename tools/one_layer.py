"""Developer tool: runs one conv pass of one layer n times (for rocprofv3 counter passes).
   python tools/one_layer.py Cin H Cout k s p {fprop|dgrad|wgrad} [n] [batch]"""
import sys
import torch
sys.path.insert(0, ".")
import gan_amd
from gan_amd import ops, _lib

Cin, H, Cout, k, s, p = (int(v) for v in sys.argv[1:7])
which = sys.argv[7]
n = int(sys.argv[8]) if len(sys.argv) > 8 else 5
B = int(sys.argv[9]) if len(sys.argv) > 9 else 256
Ho = (H + 2 * p - k) // s + 1
x = torch.randn(B, Cin, H, H, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
gy = torch.randn(B, Cout, Ho, Ho, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
wp = torch.nn.Parameter(torch.randn(Cout, Cin, k, k, device="cuda") * 0.05)
key = ops.new_weight_key()
prec = _lib.GC_PREC_BF16
fn = {"fprop": lambda: ops.conv_fprop(x, wp, None, s, p, prec, key),
      "dgrad": lambda: ops.conv_dgrad(gy, wp, None, H, H, s, p, prec, key),
      "wgrad": lambda: ops.conv_wgrad(x, gy, k, k, s, p, prec)}[which]
for _ in range(n):
    fn()
torch.cuda.synchronize()
