"""Timing of the full-attention layer (forward and backward operator entry points) on short and long sequences: the
LDS-resident kernels up to ~590 / ~300 tokens, the tiled ones beyond (DM_ATTN_TILED=1 / DM_ATTN_BWD_TILED=1 force them).
    python tools/attn_time.py"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diffusion_models_amd as dm
from diffusion_models_amd import _lib
lib = _lib.load()
DEV = "cuda:0"
def seeded(shape, seed, s=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g) * s
for (B, C, H, W) in ((64, 128, 8, 8), (64, 128, 16, 16), (64, 128, 32, 16), (64, 128, 32, 32), (8, 128, 64, 64)):
    hid = 128
    sd = {"norm": 1 + 0.25 * seeded((1, C, 1, 1), 1), "mem": seeded((2, 4, 4, 32), 2), "qkv": seeded((3 * hid, C, 1, 1), 3) / C ** 0.5,
          "out": seeded((C, hid, 1, 1), 4) / hid ** 0.5, "b": seeded((C,), 5, 0.1)}
    x = seeded((B, C, H, W), 6)
    a = [t.to(DEV).contiguous() for t in (x, sd["norm"], sd["mem"], sd["qkv"], sd["out"], sd["b"])]
    out = torch.empty((B, C, H, W), device=DEV)
    def fwd():
        _lib.check(lib.dm_op_attention(*[_lib.ptr(t) for t in a], _lib.ptr(out), B, C, H, W, 4, 32, None))
    dy = seeded((B, C, H, W), 7).to(DEV)
    dx = torch.empty_like(out); outs = [torch.empty_like(t) for t in a[1:]]
    def bwd():
        _lib.check(lib.dm_op_attention_bwd(*[_lib.ptr(t) for t in a], _lib.ptr(dy), _lib.ptr(dx), *[_lib.ptr(t) for t in outs], B, C, H, W, 4, 32, None))
    for name, f in (("forward", fwd), ("backward", bwd)):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        print(f"full attention layer B={B} C={C} {H}x{W} ({H*W} tokens) {name}: {e0.elapsed_time(e1)/5:.3f} ms")
