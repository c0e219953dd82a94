#!/usr/bin/env python3
"""Off-line sweep over random sampler configurations against the oracle's loops on identical injected noise: U-Net shape
(dim, mults, channels, image size), objective, DDPM (T = 50) and DDIM (S in 2..5, eta in {0, 0.5, 1}), self-conditioning, text
(concat / cross-attention), image condition, batch, hipGraph replay on / off.
    python tools/fuzz_sampler.py [--seed 1] [--n 12]"""
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
from oracle import sampler_oracle as so  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--n", type=int, default=12)
a = ap.parse_args()
rng = random.Random(a.seed)
torch.set_num_threads(16)
bad = 0
for it in range(a.n):
    dim = rng.choice([16, 32, 48, 64])
    mults = rng.choice([(1, 2), (1, 2, 4), (1, 1, 2), (2, 4)])
    channels = rng.choice([1, 3, 4])
    f = 2 ** (len(mults) - 1)
    H, W = f * rng.randint(1, 4), f * rng.randint(1, 4)
    B = rng.choice([1, 2, 3, 5])
    variant = rng.choice(["plain", "plain", "selfcond", "text_concat", "text_cross", "imgcond"])
    objective = rng.choice(["pred_noise", "pred_noise", "pred_x0", "pred_v"])
    use_graph = rng.random() < 0.5
    kw = dict(self_condition=variant == "selfcond", text_condition=variant.startswith("text"), use_cross_attn=variant == "text_cross",
              cond_channels=channels if variant == "imgcond" else 0)
    cfg = UnetConfig(dim=dim, dim_mults=mults, channels=channels, **kw)
    kind = rng.choice(["ddpm", "ddim"])
    T = 50 if kind == "ddpm" else 1000  # (a linear schedule needs T >= 21: beta_end = 0.02 * 1000 / T < 1)
    S = rng.randint(2, 5)
    eta = rng.choice([0.0, 0.5, 1.0])
    case = (dim, mults, channels, (H, W), B, variant, objective, kind, S if kind == "ddim" else T, eta, "graph" if use_graph else "eager")
    try:
        sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=it)
        u = dm.Unet(dim=dim, dim_mults=mults, channels=channels, device="cuda:0", **kw)
        u.load_state_dict(sd)
        g = torch.Generator().manual_seed(500 + it)
        emb = torch.randn((B, 512), generator=g) if variant.startswith("text") else None
        cond = torch.rand((B, channels, H, W), generator=g) if variant == "imgcond" else None
        common = dict(image_size=(H, W), timesteps=T, objective=objective, ddim_sampling_eta=eta, use_graph=use_graph,
                      sampling_timesteps=S if kind == "ddim" else None)
        if variant.startswith("text"):
            d = dm.TextConditionalDenoisingDiffusion(model=u, **common)
            model = lambda x, t: uo.unet_forward(sd, cfg, x, t, text_emb=emb)  # noqa: E731
            skw = {"text_emb": emb}
        elif variant == "imgcond":
            d = dm.ImageConditionalDenoisingDiffusion(u, **common)
            model = lambda x, t: uo.unet_forward(sd, cfg, x, t, cond=cond)  # noqa: E731
            skw = {"cond": cond}
        else:
            d = dm.DenoisingDiffusion(u, **common)
            model = (lambda x, t, s=None: uo.unet_forward(sd, cfg, x, t, s)) if variant == "selfcond" else (lambda x, t: uo.unet_forward(sd, cfg, x, t))
            skw = {}
        sched = dm.make_schedule(T, "linear")
        shape = (B, channels, H, W)
        seed = 900 + it
        with torch.inference_mode():
            if kind == "ddpm":
                want = so.p_sample_loop(model, sched, shape, so.NoiseStream(seed), objective=objective, self_condition=variant == "selfcond")
                got = d.p_sample_loop(shape, noise=so.NoiseStream(seed), **skw)
            else:
                want = so.ddim_sample(model, sched, shape, so.NoiseStream(seed), S, eta, objective=objective, self_condition=variant == "selfcond")
                got = d.ddim_sample(shape, noise=so.NoiseStream(seed), **skw)
        got = got[0] if isinstance(got, tuple) else got
        err = rel_l2(got.cpu(), want)
        ok = err < 1e-3
        print("OK  " if ok else "BAD ", case, f"{err:.2e}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", case, repr(e)[:200], flush=True)
print(f"seed {a.seed}: {a.n} configurations, {bad} bad")
