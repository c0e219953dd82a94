#!/usr/bin/env python3
"""Cycle anatomy of the convolution kernel: runs one U-Net forward against the DIAGNOSTIC build
(make -C diffusion-models_amd/csrc STAMPS=1) whose conv launches print per-phase s_memtime sums.
    DM_LIB=diffusion-models_amd/libdm_hip_stamps.so python tools/conv_stamps.py [--batch 256]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("DM_LIB", os.path.join(ROOT, "diffusion-models_amd", "libdm_hip_stamps.so"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--size", type=int, default=32)
args = ap.parse_args()
u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
x = torch.randn(args.batch, 3, args.size, args.size, device="cuda:0")
t = torch.full((args.batch,), 500, device="cuda:0", dtype=torch.long)
u(x, t)
torch.cuda.synchronize()
