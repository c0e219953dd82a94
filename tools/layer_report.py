#!/usr/bin/env python3
"""Per-layer-shape timing of the convolution kernels inside one U-Net forward (HIP events around every
launch, dm_profile_enable(2)).  Development aid: python tools/layer_report.py [--batch 256] [--size 32]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--size", type=int, default=32)
ap.add_argument("--iters", type=int, default=3)
args = ap.parse_args()

u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
x = torch.randn(args.batch, 3, args.size, args.size, device="cuda:0")
t = torch.full((args.batch,), 500, device="cuda:0", dtype=torch.long)
u(x, t)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    u(x, t)
e1.record()
torch.cuda.synchronize()
print(f"whole forward (eager launches): {e0.elapsed_time(e1) / 5:.3f} ms")
_lib.load().dm_profile_enable(2)
for _ in range(args.iters):
    u(x, t)
rows = _lib.profile_read()
_lib.load().dm_profile_enable(0)
rows.sort(key=lambda r: -r["total_ms"])
tot = sum(r["total_ms"] for r in rows) / args.iters
print(f"conv total {tot:.3f} ms per forward, batch {args.batch}, {args.size}x{args.size}")
print(f"{'layer':58s} {'n':>3s} {'ms':>8s} {'TF/s':>7s} {'%pk':>5s} {'GB/s':>7s} {'ms/fwd':>7s}")
for r in rows:
    n = r["launches"] // args.iters
    ms = r["total_ms"] / r["launches"]
    tf = r["total_flops"] / r["total_ms"] / 1e9
    print(f"{r['kernel']:58s} {n:3d} {ms:8.4f} {tf:7.1f} {100 * tf / 157.3:5.1f} "
          f"{r['total_bytes'] / r['total_ms'] / 1e6:7.0f} {ms * n:7.3f}")
