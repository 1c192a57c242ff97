"""Developer tool: the data-parallel graph-per-phase capture of WganGpTrainer against its eager cycle, one rank.
   python tools/ddp_graph_check.py [--pg]     (--pg: with an initialised nccl process group of one rank)"""
import os
import sys
import torch
sys.path.insert(0, ".")
import gan_amd
from gan_amd.ddp import GradReducer
from gan_amd.harness.basic_gans import WganGpTrainer, synth_inputs

if "--pg" in sys.argv:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B = 16


def make(red=True):
    tr = WganGpTrainer(B, 64, 100, 64, device="cuda:0", reducer_factory=GradReducer if red else None, capturable=True, batch_real_fake=True)
    gen = torch.Generator().manual_seed(5001)
    real, zs = synth_inputs(gen, B, 100, 64, 6)
    eps = torch.rand(5, B, 1, 1, 1, generator=gen)
    tr.set_inputs(real, zs, eps)
    return tr


a, b, c = make(), make(), make(False)
print("eager reference", flush=True)
ref = [a.cycle_eager().cpu() for _ in range(2 + 3)]     # capture() runs 2 warm-up cycles
torch.cuda.synchronize()
print("ref", [r[:3].tolist() for r in ref], flush=True)
c.capture(warmup=2)
for i in range(3):
    out = c.cycle().cpu()
    print("single graph", i, float((out - ref[2 + i]).abs().max()), out[:3].tolist(), flush=True)
print("capture", flush=True)
b.capture(warmup=2)
torch.cuda.synchronize()
print("captured", len(b.graphs), "graphs", flush=True)
b.graphs[0][0].replay(); torch.cuda.synchronize()
print("after crit0 alone:", [float(o) for o in b._outs[:3]], flush=True)
b.graphs[1][0].replay(); torch.cuda.synchronize()
print("after opt0:", [float(o) for o in b._outs[:3]], flush=True)
for g, _ in b.graphs[2:]:
    g.replay()
torch.cuda.synchronize()
print("after rest:", [round(float(o), 4) for o in b._outs], flush=True)
print("eager ref :", [round(float(o), 4) for o in ref[2]], flush=True)
print("cycle 2, graph by graph (crit0 outputs, then the just-replayed graph's outputs):", flush=True)
for gi, (g, _) in enumerate(b.graphs):
    g.replay(); torch.cuda.synchronize()
    k = (gi // 2) * 3
    print(gi, [round(float(o), 4) for o in b._outs[:3]], [round(float(o), 4) for o in b._outs[k:k + 3]], flush=True)
ref = ref[2:]
for i in range(1):
    out = b.cycle().cpu()
    torch.cuda.synchronize()
    print(i, float((out - ref[2 + i]).abs().max()), [round(float(o), 4) for o in out], flush=True)
print("ok")
