#!/usr/bin/env python3
"""Experiment: does sampling two half-batches on two independent streams (two handles, two host threads, no cross-stream
dependency) beat one batch on one stream?  Launch gaps and kernel tails of one stream could be filled by the other.
    python tools/dual_stream_test.py [--batch 256] [--steps 100]"""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--size", type=int, default=32)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--ways", type=int, default=2)
a = ap.parse_args()


def make():
    u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, device="cuda:0")
    u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=0))
    return dm.DenoisingDiffusion(u, image_size=a.size, timesteps=1000, sampling_timesteps=a.steps)


one = make()
one.sample(batch_size=a.batch, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter()
one.sample(batch_size=a.batch, seed=2)
torch.cuda.synchronize()
t_one = time.perf_counter() - t0
print(f"one stream  B={a.batch}: {1e3 * t_one / a.steps:.3f} ms/step")

ds = [make() for _ in range(a.ways)]
streams = [torch.cuda.Stream() for _ in range(a.ways)]
half = a.batch // a.ways


def run(i, seed):
    with torch.cuda.stream(streams[i]):
        ds[i].sample(batch_size=half, seed=seed, sample_offset=i * half)


for i in range(a.ways):
    run(i, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, 2)) for i in range(a.ways)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
t_two = time.perf_counter() - t0
print(f"{a.ways} streams x B={half}: {1e3 * t_two / a.steps:.3f} ms per step of the whole batch  ({t_one / t_two:.3f}x)")
