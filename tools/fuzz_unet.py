#!/usr/bin/env python3
"""Off-line sweep over random U-Net configurations (dim, dim_mults, channels, image size, batch, self-conditioning, text
conditioning): forward and, with --train, loss + every gradient, against the oracle.
    python tools/fuzz_unet.py [--seed 1] [--n 12] [--train]"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from conftest import rel_l2  # noqa: E402
from diffusion_models_amd.spec import UnetConfig  # noqa: E402
from oracle import train_oracle as to  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--n", type=int, default=12)
ap.add_argument("--train", action="store_true")
ap.add_argument("--big", action="store_true", help="maps of 4..12 times the down-sampling factor per side (several pixel tiles, long attention sequences)")
a = ap.parse_args()
rng = random.Random(a.seed)
torch.set_num_threads(16)
bad = 0
for it in range(a.n):
    dim = rng.choice([16, 24, 32, 40, 48, 64, 96])
    mults = rng.choice([(1, 2), (1, 2, 4), (1, 1, 2), (1, 2, 2, 4), (2, 4), (1, 3)])
    channels = rng.choice([1, 3, 4])
    f = 2 ** (len(mults) - 1)
    H, W = f * rng.randint(1, 4), f * rng.randint(1, 4)
    B = rng.choice([1, 2, 3, 5])
    if a.big:
        H, W = f * rng.randint(4, 12), f * rng.randint(4, 12)
        B = rng.choice([1, 2])
        dim = min(dim, 48)
    variant = rng.choice(["plain", "plain", "selfcond", "text_concat", "text_cross", "imgcond"])
    if rng.random() < 0.15:
        dim = 128
    kw = dict(self_condition=variant == "selfcond", text_condition=variant.startswith("text"), use_cross_attn=variant == "text_cross",
              cond_channels=channels if variant == "imgcond" else 0)
    if rng.random() < 0.4:  # any pattern of full / linear attention over the stages (the default: full at the last one only)
        kw["full_attn"] = tuple(rng.random() < 0.5 for _ in mults)
    if rng.random() < 0.3:  # one head count per stage (cast_tuple(attn_heads, num_stages); mid_attn takes the last)
        kw["attn_heads"] = tuple(rng.choice([1, 2, 3, 4, 6, 8]) for _ in mults)
    cfg = UnetConfig(dim=dim, dim_mults=mults, channels=channels, **kw)
    case = (dim, mults, channels, (H, W), B, variant)
    try:
        sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=it)
        u = dm.Unet(dim=dim, dim_mults=mults, channels=channels, device="cuda:0", **kw)
        u.load_state_dict(sd)
        g = torch.Generator().manual_seed(1000 + it)
        x = torch.randn((B, channels, H, W), generator=g)
        t = torch.randint(0, 1000, (B,), generator=g)
        fw = {}
        if variant == "selfcond":
            fw["x_self_cond"] = torch.randn((B, channels, H, W), generator=g)
        if variant.startswith("text"):
            fw["text_emb"] = torch.randn((B, 512), generator=g)
        if variant == "imgcond":
            fw["cond"] = torch.rand((B, channels, H, W), generator=g)
        with torch.inference_mode():
            want = uo.unet_forward(sd, cfg, x, t, **fw)
        err = rel_l2(u(x, t, **fw).cpu(), want)
        msg = f"forward {err:.2e}"
        ok = err < 1e-4
        if a.train:
            if variant.startswith("text"):
                d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=(H, W), timesteps=1000).train()
            elif variant == "imgcond":
                d = dm.ImageConditionalDenoisingDiffusion(u, image_size=(H, W), timesteps=1000).train()
            else:
                d = dm.DenoisingDiffusion(u, image_size=(H, W), timesteps=1000).train()
            noise = torch.randn((B, channels, H, W), generator=g)
            x0 = torch.rand((B, channels, H, W), generator=g) * 2 - 1
            tk = {k: v for k, v in fw.items() if k in ("text_emb", "cond")}
            okw = dict(tk)
            if variant == "selfcond":
                sc = rng.random() < 0.5
                tk["self_cond"] = sc
                okw["self_cond"] = sc
            loss = float(d.p_losses(x0, t, noise=noise, **tk))
            wl, wg = to.loss_and_grads(sd, cfg, dm.make_schedule(1000, "linear"), x0, t, noise, **okw)
            got = d.model.grads()
            worst = max((rel_l2(got[k].cpu(), wg[k]) if float(wg[k].norm()) > 0 else float(got[k].norm()), k) for k in wg)
            msg += f" loss {abs(loss - wl) / abs(wl):.1e} worst grad {worst[0]:.2e} ({worst[1]})"
            ok = ok and abs(loss - wl) <= 1e-5 * abs(wl) and worst[0] < 2e-4
        print("OK  " if ok else "BAD ", case, msg, flush=True)
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", case, repr(e)[:200], flush=True)
print(f"seed {a.seed}: {a.n} configurations, {bad} bad")
