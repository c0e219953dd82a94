#!/usr/bin/env python3
"""Off-line sweep of dm_op_block (conv3x3 + RMSNorm + scale/shift + SiLU) and dm_op_block_bwd over random shapes against
the oracle / torch autograd:  python tools/fuzz_block.py [--seed 1] [--n 150]"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from conftest import rel_l2  # noqa: E402
from diffusion_models_amd import _lib  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402
from test_hip_ops import DEV, dev, seeded  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--n", type=int, default=150)
a = ap.parse_args()
rng = random.Random(a.seed)
lib = _lib.load()
bad = 0
for it in range(a.n):
    H, W = rng.choice([(1, 1), (2, 2), (3, 3), (4, 4), (6, 6), (8, 8), (5, 9), (12, 20), (16, 16), (32, 32), (32, 16)])
    Cin = rng.choice([4, 8, 12, 24, 40, 64, 72, 128, 192, 256, 512])
    Cout = rng.choice([4, 8, 16, 32, 44, 64, 96, 128, 192, 256, 512])
    B = rng.choice([1, 2, 3, 17, 64, 130]) if H * W <= 16 else rng.choice([1, 2, 3, 5])
    ss = rng.random() < 0.6
    case = (B, Cin, Cout, H, W, ss)
    x = seeded((B, Cin, H, W), 21)
    sd = {"b.proj.weight": seeded((Cout, Cin, 3, 3), 22, (9 * Cin) ** -0.5), "b.proj.bias": seeded((Cout,), 23, 0.1),
          "b.norm.g": 1 + 0.3 * seeded((1, Cout, 1, 1), 24)}
    scale = seeded((B, Cout), 25, 0.5) if ss else None
    shift = seeded((B, Cout), 26, 0.5) if ss else None
    with torch.inference_mode():
        ref = uo.block(sd, "b", x, (scale[:, :, None, None], shift[:, :, None, None]) if ss else None)
    out = torch.empty(ref.shape, device=DEV)
    t = [dev(v) for v in (x, sd["b.proj.weight"], sd["b.proj.bias"], sd["b.norm.g"], scale, shift)]
    rc = lib.dm_op_block(_lib.ptr(t[0]), Cin, _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(t[4]), _lib.ptr(t[5]),
                         _lib.ptr(out), B, H, W, Cout, None)
    if rc:
        bad += 1
        print("FAIL", case, lib.dm_last_error().decode()[:160], flush=True)
        continue
    err = rel_l2(out.cpu(), ref)
    if not err < 2e-5:
        bad += 1
        print("MISMATCH", case, err, flush=True)
print(f"seed {a.seed}: {a.n} shapes, {bad} bad")
