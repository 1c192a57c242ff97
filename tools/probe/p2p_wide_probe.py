import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import gan_amd as gan
from test_oracle_golden import pix2pix_setup
from gan_amd.harness import pix2pix as P
gd = os.path.join("tests", "golden")
gan.set_precision(sys.argv[1])
z, rec, G, D, real, label, md, mg = pix2pix_setup(gd, lambda ch: P.Generator(ch=ch), lambda ch: P.Discriminator(ch=ch), "pix2pix_wide_trace")
G.cuda(); D.cuda(); real, label = real.cuda(), label.cuda(); md = [m.cuda() for m in md]; mg = [m.cuda() for m in mg]
print("setup done", flush=True)
with torch.no_grad():
    y = D(real, label)
torch.cuda.synchronize(); print("D fwd ok", flush=True)
with torch.no_grad():
    f = G(label, md)
torch.cuda.synchronize(); print("G fwd ok", flush=True)
d_loss = P.discriminator_phase(G, D, real, label, md)
torch.cuda.synchronize(); print("D phase ok", d_loss.item(), z["d.loss"], flush=True)
g_err, fake = P.generator_phase(G, D, real, label, 100.0, mg)
torch.cuda.synchronize(); print("G phase ok", g_err.item(), z["g.loss"], flush=True)
