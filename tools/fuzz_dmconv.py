#!/usr/bin/env python3
"""Off-line sweep over the convolution HANDLE of the sample consumer (dm_conv_create / dm_conv_forward: rectangular kernels,
strides, asymmetric padding, fused ReLU -- the InceptionV3 layer zoo) against F.conv2d on random configurations.
    python tools/fuzz_dmconv.py [--seed S] [--n N]"""
import argparse
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from diffusion_models_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--n", type=int, default=200)
a = ap.parse_args()
rng = random.Random(a.seed)
torch.set_num_threads(16)
lib = _lib.load()
DEV = "cuda:0"
bad = 0
for it in range(a.n):
    cin = rng.choice([1, 3, 4, 8, 16, 24, 32, 48, 64, 80, 96, 128, 160, 192, 288, 384])
    cout = rng.choice([8, 16, 32, 48, 64, 96, 128, 192, 320, 384])
    kh, kw = rng.choice([1, 3, 5, 7]), rng.choice([1, 3, 5, 7])
    stride = rng.choice([1, 1, 2])
    ph, pw = rng.choice([0, kh // 2]), rng.choice([0, kw // 2])
    H, W = rng.randint(max(kh - 2 * ph, 1), 40), rng.randint(max(kw - 2 * pw, 1), 40)
    B = rng.choice([1, 2, 5])
    relu = rng.random() < 0.5
    case = (cin, cout, (kh, kw), stride, (ph, pw), (H, W), B, relu)
    g = torch.Generator().manual_seed(100 + it)
    w = torch.randn(cout, cin, kh, kw, generator=g) * (2.0 / (cin * kh * kw)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = torch.randn(B, cin, H, W, generator=g)
    ref = F.conv2d(x, w, b, stride=stride, padding=(ph, pw))
    if relu:
        ref = F.relu(ref)
    try:
        h = C.c_void_p()
        _lib.check(lib.dm_conv_create(w.data_ptr(), b.data_ptr(), cout, cin, kh, kw, stride, ph, pw, int(relu), 0, C.byref(h)))
        xin = x.permute(0, 2, 3, 1).contiguous().to(DEV)
        y = torch.empty((B, ref.shape[2], ref.shape[3], cout), device=DEV)
        _lib.check(lib.dm_conv_forward(h, _lib.ptr(xin), 0, B, H, W, _lib.ptr(y), None))
        torch.cuda.synchronize()
        lib.dm_conv_destroy(h)
        got = y.cpu().permute(0, 3, 1, 2)
        err = float((got - ref).norm() / ref.norm().clamp_min(1e-20))
        ok = err < 2e-5
        print(("OK  " if ok else "FAIL"), case, f"{err:.2e}")
        bad += 0 if ok else 1
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", case, repr(e)[:300])
print(f"seed {a.seed}: {a.n} configurations, {bad} bad")
