#!/usr/bin/env python3
"""Time of one training step's loss + backward (dm_unet_loss_backward) on the 32x32 U-Net, synthetic data.
    python tools/train_time.py [--batch 64] [--steps 10] [--size 32]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--warmup", type=int, default=10, help="untimed full iterations first (the first ones run at ramping clocks and size the workspace; profiling scripts pass 1)")
ap.add_argument("--size", type=int, default=32)
ap.add_argument("--dropout", type=float, default=0.0)
ap.add_argument("--full-only", action="store_true", help="only full Trainer.train iterations: --warmup + --steps (profiling: --warmup 1)")
ap.add_argument("--host", action="store_true", help="also print the host time to ENQUEUE an iteration")
args = ap.parse_args()
u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, dropout=args.dropout, device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
d = dm.DenoisingDiffusion(u, image_size=args.size, timesteps=1000).train()
img = torch.rand(args.batch, 3, args.size, args.size, device="cuda:0")
torch.manual_seed(0)
if not args.full_only:
    d(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = d(img, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"loss+backward B={args.batch} {args.size}x{args.size}: {1e3 * dt:.2f} ms/step  {args.batch / dt:.1f} images/s  (loss {float(loss):.4f})")
ema = dm.EMA(d, beta=0.995, update_every=10)
for _ in range(max(1, args.warmup)):
    dm.train_step(d, [img], lr=2e-4, ema=ema)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    loss, norm = dm.train_step(d, [img], lr=2e-4, ema=ema, sync=False)
t_host = (time.perf_counter() - t0) / args.steps
torch.cuda.synchronize()
loss, norm = float(loss), float(norm)
dt = (time.perf_counter() - t0) / args.steps
print(f"full iteration (loss+backward, clip, Adam, device re-pack, EMA) B={args.batch}: {1e3 * dt:.2f} ms/step  "
      f"{args.batch / dt:.1f} images/s  (loss {loss:.4f}, grad norm {norm:.3f})")
if args.host:
    print(f"host time to enqueue one iteration: {1e3 * t_host:.2f} ms")
