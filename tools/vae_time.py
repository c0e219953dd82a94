#!/usr/bin/env python3
"""ms per VQModel.decode at the config-4 shape (B=128, 4x32x32 latents -> 3x64x64), development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusion_models_amd as dm
from diffusion_models_amd.spec import DecoderConfig
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
cfg = DecoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4, embed_dim=4)
vae = dm.VQDecoder(dict(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4), embed_dim=4, device="cuda:0")
vae.load_state_dict(dm.synth_state_dict(dm.decoder_param_spec(cfg), salt=4))
z = torch.randn(B, 4, 32, 32, device="cuda:0")
vae.decode(z); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): vae.decode(z)
torch.cuda.synchronize()
print(f"decode B={B}: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms")
