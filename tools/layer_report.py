#!/usr/bin/env python3
"""Per-layer-shape timing of the convolution kernels inside one U-Net forward (HIP events around every
launch, dm_profile_enable(2)).  Development aid: python tools/layer_report.py [--batch 256] [--size 32]

--bounds adds, per launch shape, what bounds it:
  t_mfma  = the f32-MFMA FLOPs the kernel EXECUTES (F(2x2): 16/36, F(4x4) and the upsample algorithm: 9/36 of the direct count)
            / (157.3 TFLOP/s x the share of the 256 CUs that has a workgroup)
  t_chain = rounds x chunks x MFMA cycles per chunk / 2.4 GHz: the MFMA issue time of ONE workgroup (all four SIMDs busy),
            times the rounds the grid needs (workgroups / (256 CUs x workgroups per CU)) -- what the launch costs when nothing
            but matrix instructions were issued
  t_wts   = bytes of packed weights all workgroups pull through L2 -> CU / (64 B/clk x 2.4 GHz x busy CUs)
  fill    = workgroups / 256
and time / max(bound).  The gap to the bound is the per-workgroup fixed cost (first windows from L2 / HBM, the accumulator
exchange, output transform, norm / SiLU, stores: 8-10 us for the 3x3 kernels) that one round of workgroups cannot hide."""
import re
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
ap.add_argument("--bounds", action="store_true")
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


if args.bounds:
    PEAK, CLK, L2B = 157.3e12, 2.4e9, 64.0
    pat = re.compile(r"^(wino4|wino|upwino|pw)<(\d)> \S+ (?:s1 |up )?(\d+)(?:\+(\d+))?->(\d+) @(\d+)x(\d+)( s2d)? e\d+ k(\d+) g(\d+)(?: [rq](\d))?")
    print()
    print(f"{'layer':58s} {'n':>3s} {'us':>6s} {'t_mfma':>6s} {'t_chain':>7s} {'t_wts':>6s} {'fill':>5s} {'x bound':>7s}")
    worst = []
    for r in rows:
        m = pat.match(r["kernel"])
        if not m:
            continue
        kind, var, c0, c1, cout, ho, wo, s2d, k, g, rt = m.groups()
        c0, c1, cout, k, g = int(c0), int(c1 or 0), int(cout), int(k), int(g)
        cin = (4 * c0 if s2d else c0 + c1)
        us = 1e3 * r["total_ms"] / r["launches"]
        direct = r["total_flops"] / r["launches"]
        # executed share, chunk size (input channels), MFMA cycles per chunk and workgroup, transformed-weight factor, WGs per CU
        share, ck, cyc, wfac, per_cu = {"wino": (16 / 36, 8, 2048, 16 / 9, 2), "wino4": (0.25, 8, 2304, 4.0, 1),
                                        "upwino": (0.25, 8, 2304, 1.0, 1), "pw": (1.0, 16, 2048, 1.0, 2)}[kind]
        taps = 1 if kind == "pw" else 9
        if kind == "pw" and rt:  # 16-pixel row tiles per wave (4, 2, 1): the chain and the weight reuse scale with it
            cyc = cyc * int(rt) // 4
        if kind == "wino" and rt:  # 32-cout blocks per workgroup (2, 1)
            cyc = cyc * int(rt) // 2
        busy = min(g, 256)
        t_mfma = 1e6 * direct * share / (PEAK * busy / 256)
        rounds = -(-g // (256 * per_cu))
        chunks = -(-(cin // ck) // k)
        t_chain = 1e6 * rounds * chunks * cyc / CLK
        nc = 32 * int(rt) if (kind == "wino" and rt) else 64
        wbytes = g * (cin / k) * taps * wfac * nc * 4.0   # every workgroup streams its cout slice of its K range
        t_wts = 1e6 * wbytes / (L2B * CLK * busy)
        bound = max(t_mfma, t_chain, t_wts)
        n = r["launches"] // args.iters
        print(f"{r['kernel']:58s} {n:3d} {us:6.1f} {t_mfma:6.1f} {t_chain:7.1f} {t_wts:6.1f} {g / 256:5.2f} {us / bound:7.2f}")
        worst.append((n * (us - bound), r["kernel"], n, us, bound))
    worst.sort(reverse=True)
    print()
    print("launch shapes by time above their bound (n x (us - bound)):")
    for gap, name, n, us, bound in worst[:8]:
        print(f"  {gap:7.1f} us  {name}  ({n} x {us:.1f} us, bound {bound:.1f})")
