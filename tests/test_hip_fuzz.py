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


def _block_cases():
    rng = random.Random(4242)
    out = []
    for _ in range(8):
        size = rng.choice([(16, 16), (32, 16), (8, 8), (4, 4), (12, 20), (6, 6)])
        cout = rng.choice([64, 64, 128, 256])
        cin = 8 * rng.randint(2, 24)
        need = 220 * 256 // max(1, cout // 64)
        b = max(2, min(96, -(-need // (size[0] * size[1]))))
        out.append((b + rng.randint(0, 2), cin, cout, size[0], size[1], rng.random() < 0.6))
    return out


BLOCK_RANDOM = _block_cases()


@pytest.mark.parametrize("case", BLOCK_RANDOM, ids=[str(c) for c in BLOCK_RANDOM])
def test_block_random_shapes(case):
    """Block.forward (conv3x3 + RMSNorm + scale / shift + SiLU) on drawn shapes: fused epilogues of the Winograd kernels
    (Cout = 64) and the landing pass (several cout tiles)."""
    import ctypes  # noqa: F401

    from diffusion_models_amd import _lib
    from oracle import unet_oracle as uo
    from test_hip_ops import DEV, dev

    B, Cin, Cout, H, W, ss = case
    x = seeded((B, Cin, H, W), 21)
    sd = {
        "b.proj.weight": seeded((Cout, Cin, 3, 3), 22, (9 * Cin) ** -0.5),
        "b.proj.bias": seeded((Cout,), 23, 0.1),
        "b.norm.g": 1 + 0.3 * seeded((1, Cout, 1, 1), 24),
    }
    scale = seeded((B, Cout), 25, 0.5) if ss else None
    shift = seeded((B, Cout), 26, 0.5) if ss else None
    ref = uo.block(sd, "b", x, (scale[:, :, None, None], shift[:, :, None, None]) if ss else None)
    out = torch.empty(ref.shape, device=DEV)
    a = [dev(t) for t in (x, sd["b.proj.weight"], sd["b.proj.bias"], sd["b.norm.g"], scale, shift)]
    _lib.check(_lib.load().dm_op_block(_lib.ptr(a[0]), Cin, _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]),
                                       _lib.ptr(a[4]), _lib.ptr(a[5]), _lib.ptr(out), B, H, W, Cout, None))
    assert rel_l2(out.cpu(), ref) < TOL
