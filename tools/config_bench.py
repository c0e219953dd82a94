#!/usr/bin/env python3
"""Per-GPU throughput of the other BASELINE configs (3: 64x64 U-Net, 4: latent 4x32x32 + VAE decode, 5: text-conditional
64x64) on ONE MI355X: graph-replayed sampler, synthetic weights, a short step count (the per-step cost does not depend on
the schedule length).  Development aid; bench.py stays on configs[1].
    python tools/config_bench.py [--steps 20]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd.spec import DecoderConfig  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
args = ap.parse_args()
DEV = "cuda:0"


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def unet(**kw):
    u = dm.Unet(device=DEV, **kw)
    u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
    return u


S = args.steps
# config 3: 64x64, 32 images per GPU (256 over 8 GPUs)
u = unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3)
d = dm.DenoisingDiffusion(u, image_size=64, timesteps=1000, sampling_timesteps=S)
for B in (32, 8):
    dt = timed(lambda: d.sample(batch_size=B, seed=1))
    print(f"config3 64x64 U-Net   B={B:4d}: {1e3 * dt / S:7.3f} ms/step  {B * S / dt:9.1f} image-steps/s  "
          f"DDPM-1000: {B * S / dt / 1000:.2f} img/s/GPU")

# config 4: latent 4x32x32, B=128, DDIM-200, + VAE decode to 3x64x64
u4 = unet(dim=64, dim_mults=(1, 2, 4, 8), channels=4)
cfg = DecoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4,
                    embed_dim=4)
vae = dm.VQDecoder(dict(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64,
                        z_channels=4), embed_dim=4, device=DEV)
vae.load_state_dict(dm.synth_state_dict(dm.decoder_param_spec(cfg), salt=4))
ld = dm.LatentDiffusion(u4, vae, latent_shape=(4, 32, 32), timesteps=1000, sampling_timesteps=S)
B = 128
dt_loop = timed(lambda: ld.ddim_sample((B, 4, 32, 32), seed=1))
z = ld.ddim_sample((B, 4, 32, 32), seed=1)
dt_dec = timed(lambda: vae.decode(z))
print(f"config4 latent 4x32x32 B={B:4d}: {1e3 * dt_loop / S:7.3f} ms/step, decode {1e3 * dt_dec:.2f} ms  "
      f"DDIM-200 + decode: {B / (200 * dt_loop / S + dt_dec):.2f} img/s")

# config 5: text-conditional 64x64 (cross-attention at the bottleneck), 32 per GPU, DDIM-100
ut = unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, text_condition=True, use_cross_attn=True)
dt5 = dm.TextConditionalDenoisingDiffusion(model=ut, image_size=64, timesteps=1000, sampling_timesteps=S)
B = 32
emb = torch.randn(B, 512, device=DEV)
dt = timed(lambda: dt5.sample(batch_size=B, text_emb=emb, seed=1))
print(f"config5 text 64x64      B={B:4d}: {1e3 * dt / S:7.3f} ms/step  DDIM-100: {B / (100 * dt / S):.2f} img/s/GPU")
