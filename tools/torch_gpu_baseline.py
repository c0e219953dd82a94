#!/usr/bin/env python3
"""What stock PyTorch-ROCm (MIOpen / rocBLAS, eager) does with the same work on the same GPU: the ORACLE's U-Net (plain
torch ops = what the reference's modules call) moved to cuda:0, fp32.  A measurement aid for DESIGN.md, not a product path:
    python tools/torch_gpu_baseline.py [--batch 256] [--train-batch 64]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd.spec import UnetConfig  # noqa: E402
from oracle import sampler_oracle as so  # noqa: E402
from oracle import train_oracle as to  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--train-batch", type=int, default=64)
ap.add_argument("--size", type=int, default=32)
args = ap.parse_args()
dev = "cuda:0"
torch.backends.cuda.matmul.allow_tf32 = False
torch.backends.cudnn.allow_tf32 = False
cfg = UnetConfig()
sd = {k: v.to(dev) for k, v in dm.synth_state_dict(dm.unet_param_spec(cfg), salt=0).items()}
sched = {k: v.to(dev) for k, v in dm.make_schedule(1000, "linear").items()}
x = torch.randn(args.batch, 3, args.size, args.size, device=dev)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


bt = torch.full((args.batch,), 500, device=dev, dtype=torch.long)
with torch.inference_mode():
    def step():  # p_sample (DD/denoising_diffusion.py:638-645) at t = 500, on the device
        global x
        eps = uo.unet_forward(sd, cfg, x, bt)
        x0 = (sched["sqrt_recip_alphas_cumprod"][500] * x - sched["sqrt_recipm1_alphas_cumprod"][500] * eps).clamp(-1, 1)
        mean = sched["posterior_mean_coef1"][500] * x0 + sched["posterior_mean_coef2"][500] * x
        x = mean + (0.5 * sched["posterior_log_variance_clipped"][500]).exp() * torch.randn_like(x)
    dt = timed(step, 10)
print(f"torch-ROCm eager, oracle p_sample (DDPM step) B={args.batch} {args.size}x{args.size}: {1e3 * dt:.2f} ms/step "
      f"-> {args.batch / (1000 * dt):.2f} images/s at DDPM-1000")

params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
opt = torch.optim.Adam(list(params.values()), lr=2e-4, betas=(0.9, 0.99))
B = args.train_batch
xs = torch.rand(B, 3, args.size, args.size, device=dev) * 2 - 1
tt = torch.randint(0, 1000, (B,), device=dev)
nz = torch.randn_like(xs)


def it():
    opt.zero_grad()
    loss = to.p_losses(params, cfg, sched, xs, tt, nz)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)
    opt.step()


dt = timed(it, 5)
print(f"torch-ROCm eager, oracle training iteration (autograd + clip + Adam) B={B}: {1e3 * dt:.2f} ms/iteration "
      f"-> {B / dt:.1f} images/s")
