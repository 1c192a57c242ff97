"""Developer probe: the framework (non-library) operators of one eager training step of a model family, by device time,
with their input shapes:   python tools/probe/family_glue_ops.py biggan128_b256"""
import sys
sys.path.insert(0, ".")
import torch
from torch.profiler import profile, ProfilerActivity
import gan_amd
from gan_amd.harness import steps

name = sys.argv[1]
step = steps.FAMILIES[name]("cuda:0")
for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if not e.key.startswith("aten::"):
        continue
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("framework operators with device time: %.3f ms in one step" % (tot / 1e3))
for dt, n, k, sh in rows[:60]:
    print("%9.1f us  n=%4d  %-34s %s" % (dt, n, k[:34], sh))
