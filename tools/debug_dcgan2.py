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
for kind in ("mine_x3", "torch_gpu"):
    gan_amd.set_precision("fp32x3")
    if kind == "mine_x3":
        G, D = H.Generator(100, 8, 3, 64), H.Discriminator(3, 8, 64)
    else:
        G, D = R.Generator(100, 8, 3, resolution=64), R.Discriminator(3, 8, resolution=64)
    load(G, "init.G."); load(D, "init.D."); G.cuda(); D.cuda()
    real = torch.from_numpy(z["real"]).cuda(); noise = torch.from_numpy(z["z"]).cuda()
    recs = []
    for tag, net in (("G", G), ("D", D)):
        for name, mod in net.named_modules():
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                def hook(m, i, o, name=name, tag=tag):
                    r = {"tag": tag, "name": name, "y": o.detach().clone(), "x": i[0].detach().clone()}
                    recs.append(r)
                    if o.requires_grad: o.register_hook(lambda g, r=r: r.__setitem__("gy", g.detach().clone()))
                mod.register_forward_hook(hook)
    H.dcgan_discriminator_phase(G, D, real, noise)
    print("==", kind)
    cnt = {"G": 0, "D": 0}
    for r in recs:
        key = f"trace.{r['tag']}.call{cnt[r['tag']]:02d}.{r['name']}"; cnt[r['tag']] += 1
        print(key, "x", f"{rel(r['x'], z[key + '.x']):.2e}", "y", f"{rel(r['y'], z[key + '.y']):.2e}", "gy", f"{rel(r['gy'], z[key + '.gy']):.2e}" if (key + ".gy" in z.files and "gy" in r) else "")
    for k, p in D.named_parameters():
        print("   grad", k, f"{rel(p.grad, z['dphase.gradD.' + k]):.2e}")
