import json, os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import gan_amd
from gan_amd.harness import basic_gans as H
from oracle import ref_models as R
rec = json.load(open("tests/golden/wgangp_losses.json")); cfg = rec["config"]
for prec, adt in (("fp32x3", torch.float32), ("bf16", torch.float32), ("bf16", torch.bfloat16)):
    gan_amd.set_precision(prec)
    tr = H.WganGpTrainer(cfg["B"], cfg["hidden_dim"], cfg["nz"], cfg["res"], lr=cfg["lr"], betas=tuple(cfg["betas"]), seed=cfg["seed"],
                         capturable=False, batch_real_fake=True, fused_adam=True, activation_dtype=adt)
    torch.manual_seed(cfg["seed"])
    G = R.Generator(cfg["nz"], cfg["hidden_dim"], 3, resolution=cfg["res"]); G.apply(R.weights_init)
    D = R.Discriminator(3, cfg["hidden_dim"], resolution=cfg["res"], is_wgan=True); D.apply(R.weights_init)
    tr.net_G.load_state_dict(G.state_dict()); tr.net_D.load_state_dict(D.state_dict())
    gen = torch.Generator().manual_seed(cfg["data_seed"])
    real, zs = R.synth_inputs(gen, cfg["B"], cfg["nz"], cfg["res"], 6)
    eps = torch.rand(5, cfg["B"], 1, 1, 1, generator=gen)
    tr.set_inputs(real, zs, eps)
    out = tr.cycle_eager().cpu().numpy()
    want = rec["cycles"][0]
    devs = []
    for it in range(5):
        w = want["critic"][it]
        d_x, d_g, gp = out[3 * it:3 * it + 3]
        devs.append((abs(d_x - w["D_x"]), abs(d_g - w["D_G_z1"]), abs(gp - w["gp"]) / abs(w["gp"])))
    print(prec, adt, "max |dD_x| %.4f  max |dD_G_z1| %.4f  max rel dgp %.4f  g_err dev %.4f" % (max(d[0] for d in devs), max(d[1] for d in devs), max(d[2] for d in devs), abs(out[15] - want["generator"]["g_error"]) if isinstance(want.get("generator"), dict) else -1))
    print("   D_x", [round(float(out[3*i]),4) for i in range(5)], [round(want["critic"][i]["D_x"],4) for i in range(5)])
