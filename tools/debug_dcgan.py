import sys, numpy as np, torch
sys.path.insert(0, ".")
import gan_amd
from gan_amd.harness import basic_gans as H
from oracle import ref_models as R
z = np.load("tests/golden/dcgan_trace.npz")
def rel(a, b):
    a = np.asarray(a.detach().cpu().numpy() if torch.is_tensor(a) else a, np.float64); b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
def load(net, pre):
    net.load_state_dict({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)})
for kind in ("mine_x3", "torch_gpu", "torch_gpu_cl"):
    gan_amd.set_precision("fp32x3")
    if kind == "mine_x3":
        G, D = H.Generator(100, 8, 3, 64), H.Discriminator(3, 8, 64)
    else:
        G, D = R.Generator(100, 8, 3, resolution=64), R.Discriminator(3, 8, resolution=64)
    load(G, "init.G."); load(D, "init.D."); G.cuda(); D.cuda()
    real = torch.from_numpy(z["real"]).cuda(); noise = torch.from_numpy(z["z"]).cuda()
    if kind == "torch_gpu_cl":
        real = real.contiguous(memory_format=torch.channels_last)
    recs = {}
    for name, mod in D.named_modules():
        if isinstance(mod, torch.nn.Conv2d):
            def hook(m, i, o, name=name):
                if name in recs: return
                recs[name] = {"y": o.detach().clone()}
                o.register_hook(lambda g, name=name: recs[name].__setitem__("gy", g.detach()))
            mod.register_forward_hook(hook)
    out = D(real).view(-1)
    loss = torch.nn.functional.binary_cross_entropy(out, torch.ones_like(out)); loss.backward()
    print("==", kind)
    for i, name in enumerate(recs):
        key = f"trace.D.call{i:02d}.{name}"
        if i == 1:
            for k, p in D.named_parameters():
                print("   grad", k, f"{rel(p.grad, z['dphase.gradD.' + k]):.2e}" if False else "")
        print(name, "y", f"{rel(recs[name]['y'], z[key + '.y']):.2e}", "gy", f"{rel(recs[name]['gy'], z[key + '.gy']):.2e}" if key + ".gy" in z.files else "")
