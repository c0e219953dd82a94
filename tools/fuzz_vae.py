#!/usr/bin/env python3
"""Off-line sweep over random VQModel configurations (ch, ch_mult, num_res_blocks, attention resolutions, z channels,
resolution, batch): decode, encode_to_prequant and the quantised encode against the oracle.
    python tools/fuzz_vae.py [--seed 1] [--n 10]"""
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
from diffusion_models_amd.spec import DecoderConfig, EncoderConfig, encoder_param_spec  # noqa: E402
from oracle import vae_oracle as vo  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--n", type=int, default=10)
a = ap.parse_args()
rng = random.Random(a.seed)
torch.set_num_threads(16)
bad = 0
for it in range(a.n):
    ch = rng.choice([32, 64, 96])
    ch_mult = rng.choice([(1, 2), (1, 2, 4), (1, 1, 2), (1, 2, 2, 4), (1,)])
    nrb = rng.choice([1, 2])
    res = rng.choice([16, 32, 64]) if len(ch_mult) < 4 else rng.choice([32, 64])
    zc = rng.choice([3, 4])
    zres = res // 2 ** (len(ch_mult) - 1)
    attn = rng.choice([(), (), (zres,)])
    n_embed = rng.choice([256, 1024])
    B = rng.choice([1, 2, 3])
    case = (ch, ch_mult, nrb, res, zc, attn, n_embed, B)
    try:
        common = dict(ch=ch, ch_mult=ch_mult, num_res_blocks=nrb, resolution=res, z_channels=zc, embed_dim=zc, attn_resolutions=attn)
        ecfg, dcfg = EncoderConfig(n_embed=n_embed, **common), DecoderConfig(**common)
        sd = dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=50 + it)
        vae = dm.VQModel(dict(out_ch=3, in_channels=3, double_z=False, **{k: v for k, v in common.items() if k != "embed_dim"}),
                         n_embed=n_embed, embed_dim=zc, device="cuda:0")
        vae.load_state_dict(sd)
        g = torch.Generator().manual_seed(it)
        z = torch.randn((B, zc, zres, zres), generator=g)
        img = torch.rand((B, 3, res, res), generator=g) * 2 - 1
        with torch.inference_mode():
            e_dec = rel_l2(vae.decode(z).cpu(), vo.vq_decode(sd, dcfg, z))
            e_pre = rel_l2(vae.encode_to_prequant(img).cpu(), vo.vq_encode_to_prequant(sd, ecfg, img))
            qz, _, (_, _, idx) = vae.encode(img)
            wz, widx = vo.vq_encode(sd, ecfg, img)
        same = float((idx.cpu().reshape(-1) == widx.reshape(-1)).float().mean())
        ok = e_dec < 1e-4 and e_pre < 1e-4 and same > 0.99
        print("OK  " if ok else "BAD ", case, f"decode {e_dec:.2e} prequant {e_pre:.2e} same codes {same:.4f}", flush=True)
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", case, repr(e)[:200], flush=True)
print(f"seed {a.seed}: {a.n} configurations, {bad} bad")
