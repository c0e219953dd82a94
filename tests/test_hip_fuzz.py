"""Seeded random shapes through dm_op_conv2d against torch's CPU convolution: shapes nobody wrote down by hand, drawn
so that every dispatch target is hit with its PRODUCTION thresholds -- F(4x4,3x3) (>= 200 workgroups, >= 12 chunks),
the upsample algorithm (>= 128 workgroups), the 1x1 GEMM kernel, F(2x2,3x3) and the direct kernel."""
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2
from test_hip_ops import hip_conv, seeded

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _cases():
    rng = random.Random(20261004)
    out = []
    # F(4x4,3x3): 16k x 16m / 8x8 / 4x4 images, C % 8 == 0, Cout % 64 == 0, >= 12 chunks of 8, >= 200 workgroups
    for _ in range(5):
        size = rng.choice([(16, 16), (32, 16), (16, 48), (8, 8), (4, 4)])
        cout = rng.choice([64, 128])
        c0 = 8 * rng.randint(6, 16)
        c1 = 8 * rng.randint(0, 6)
        if c0 + c1 < 96:
            c1 = 96 - c0
        px_per_wg = 256
        need = 200 * px_per_wg // (cout // 64)
        b = max(1, -(-need // (size[0] * size[1])))
        out.append((b + rng.randint(0, 3), c0, c1, size[0], size[1], cout, 3, 1, False, rng.random() < 0.7, rng.random() < 0.5))
    # upsample algorithm: source 4x4 or multiples of 8, C % 8 == 0, single source, >= 128 workgroups, >= 8 chunks
    for _ in range(4):
        src = rng.choice([(4, 4), (8, 8), (8, 16), (16, 8)])
        cout = rng.choice([64, 128])
        c0 = 8 * rng.randint(8, 20)
        per_wg = 64
        need = 128 * per_wg // (cout // 64)
        b = max(1, -(-need // (src[0] * src[1])))
        out.append((b + rng.randint(0, 2), c0, 0, src[0], src[1], cout, 3, 1, True, rng.random() < 0.7, False))
    # 1x1 GEMM kernel: C % 16 == 0, Cout % 64 == 0, any pixel count
    for _ in range(6):
        c0 = 16 * rng.randint(1, 24)
        c1 = 16 * rng.randint(0, 8)
        cout = 64 * rng.randint(1, 6)
        h, w = rng.randint(1, 9), rng.randint(1, 9)
        out.append((rng.randint(1, 40), c0, c1, h, w, cout, 1, 0, False, rng.random() < 0.7, rng.random() < 0.5))
    # F(2x2,3x3) / direct: anything goes
    for _ in range(10):
        k = rng.choice([1, 3, 3, 3, 5, 7])
        c0 = rng.choice([3, 4, 6, 8, 24, 40, 64, 72])
        c1 = rng.choice([0, 0, 8, 16]) if c0 % 4 == 0 else 0
        cout = rng.choice([8, 16, 48, 64, 128])
        h, w = rng.randint(max(2, k // 2 + 1), 20), rng.randint(max(2, k // 2 + 1), 20)
        out.append((rng.randint(1, 6), c0, c1, h, w, cout, k, k // 2, False, rng.random() < 0.7, rng.random() < 0.4))
    return out


CASES = _cases()


@pytest.mark.parametrize("case", CASES, ids=[str(c) for c in CASES])
def test_conv2d_random_shapes(case):
    B, C0, C1, H, W, Cout, k, pad, up2, bias, residual = case
    x0 = seeded((B, C0, H, W), 11)
    x1 = seeded((B, C1, H, W), 12) if C1 else None
    w = seeded((Cout, C0 + C1, k, k), 13, (C0 + C1) ** -0.5 / k)
    b = seeded((Cout,), 14) if bias else None
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    if up2:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = F.conv2d(xin, w, b, padding=pad)
    res = seeded(ref.shape, 15) if residual else None
    if residual:
        ref = ref + res
    got = hip_conv(x0, w, b, x1, res, pad, up2)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < TOL
