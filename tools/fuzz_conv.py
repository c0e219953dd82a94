#!/usr/bin/env python3
"""Off-line sweep of dm_op_conv2d over random shapes (forward; the backward sweep is DM_TEST_SEED / DM_TEST_SHAPES of
tests/test_hip_train_ops.py): prints every shape that fails to launch or misses torch's CPU convolution by more than 2e-5.
    python tools/fuzz_conv.py [--seed 1] [--n 300]"""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from conftest import rel_l2  # noqa: E402
from test_hip_ops import hip_conv, seeded  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--n", type=int, default=300)
a = ap.parse_args()
rng = random.Random(a.seed)
bad = 0
for it in range(a.n):
    k = rng.choice([1, 1, 3, 3, 3, 5, 7])
    h, w = rng.choice([(1, 1), (2, 2), (1, 3), (3, 3), (4, 4), (5, 7), (8, 8), (6, 10), (16, 16), (12, 20), (32, 32), (33, 9),
                       (64, 64), (10, 34)])
    c0 = rng.choice([3, 4, 6, 8, 12, 24, 40, 44, 64, 72, 100, 128, 192, 256, 320, 512])
    c1 = rng.choice([0, 0, 0, 8, 16, 64, 100]) if c0 % 4 == 0 else 0
    cout = rng.choice([3, 4, 8, 16, 44, 48, 64, 100, 128, 192, 256, 384, 512])
    if k > 3 and h * w <= 9:
        k = 3  # kernels far larger than the map are refused by design (a tile's window would be 25-49x the tile)
    up2 = k == 3 and c1 == 0 and rng.random() < 0.2
    b = rng.choice([1, 2, 3, 7, 17, 64, 300]) if h * w <= 16 else rng.choice([1, 2, 3, 5])
    bias, residual = rng.random() < 0.7, rng.random() < 0.4
    case = (b, c0, c1, h, w, cout, k, k // 2, up2, bias, residual)
    x0 = seeded((b, c0, h, w), 11)
    x1 = seeded((b, c1, h, w), 12) if c1 else None
    wt = seeded((cout, c0 + c1, k, k), 13, (c0 + c1) ** -0.5 / k)
    bs = seeded((cout,), 14) if bias else None
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    if up2:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = F.conv2d(xin, wt, bs, padding=k // 2)
    res = seeded(ref.shape, 15) if residual else None
    if residual:
        ref = ref + res
    try:
        got = hip_conv(x0, wt, bs, x1, res, k // 2, up2)
        err = rel_l2(got, ref)
        if not err < 2e-5:
            bad += 1
            print("MISMATCH", case, err, flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", case, str(e)[:160], flush=True)
print(f"seed {a.seed}: {a.n} shapes, {bad} bad")
