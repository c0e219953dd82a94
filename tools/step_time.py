#!/usr/bin/env python3
"""ms per denoise step of the hipGraph-replayed sampler for a given batch / image size (development aid)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diffusion_models_amd as dm
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--no-graph", action="store_true")
a = ap.parse_args()
u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, device="cuda:0")
u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
d = dm.DenoisingDiffusion(u, image_size=a.size, timesteps=1000, sampling_timesteps=a.steps, use_graph=not a.no_graph)
d.sample(batch_size=a.batch, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3):
    d.sample(batch_size=a.batch, seed=2 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"B={a.batch} {a.size}x{a.size} graph={not a.no_graph}: {1e3*dt/a.steps:.3f} ms/step, {a.batch/dt:.1f} img/s (DDIM-{a.steps})")
