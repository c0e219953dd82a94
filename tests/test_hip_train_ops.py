"""GPU parity of the single BACKWARD operators of the training step (dm_op_*_bwd, through the C ABI) against torch
autograd on the CPU through the oracle's restatement of the reference modules (plain fp32 torch ops), on seeded inputs.

The whole-model gradient test against the reference's own autograd is tests/test_hip_train.py; these cases cover shapes
the U-Net does not produce: odd image sizes (ragged pixel blocks of the weight-gradient kernel), two-source inputs,
upsampled sources, channel counts that are not multiples of 64, the dispatch of the input-gradient convolution to the
Winograd / 1x1 GEMM / direct kernels.

Tolerance: rel-L2 <= 5e-5 per gradient tensor (fp32, different summation order; measured ~1e-6)."""
import pytest
import torch
import torch.nn.functional as F

from diffusion_models_amd import _lib
from oracle import unet_oracle as uo

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 5e-5
DEV = "cuda:0"


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def dev(t):
    return None if t is None else t.to(DEV).contiguous()


def empty_like_dev(t):
    return None if t is None else torch.empty(t.shape, device=DEV)


CONV_BWD_CASES = [
    # (B, C0, C1, H, W, Cout, k, up2)
    (2, 64, 0, 32, 32, 64, 3, False),     # Block conv @32^2
    (3, 64, 64, 16, 16, 64, 3, False),    # two sources (the up path's torch.cat)
    (2, 128, 64, 8, 8, 128, 3, False),
    (5, 256, 0, 4, 4, 256, 3, False),     # 16-pixel blocks, ragged batch
    (1, 512, 256, 4, 4, 512, 3, False),   # deepest layer
    (2, 32, 0, 8, 8, 16, 3, True),        # nearest x2 in front of the conv, thin channels
    (2, 128, 0, 8, 8, 64, 3, True),       # Upsample 128 -> 64
    (2, 20, 12, 7, 9, 36, 3, False),      # odd sizes, channel counts that are only multiples of 4
    (1, 64, 0, 64, 64, 64, 3, False),     # 64-wide rows: one row per pixel block
    (2, 64, 0, 16, 16, 384, 1, False),    # to_qkv 1x1
    (2, 128, 64, 8, 8, 64, 1, False),     # res_conv over a concatenation
    (3, 24, 0, 5, 6, 40, 1, False),       # 1x1, odd everything
]


@pytest.mark.parametrize("case", CONV_BWD_CASES)
def test_conv2d_bwd(case):
    B, C0, C1, H, W, Cout, k, up2 = case
    pad = k // 2
    x0 = seeded((B, C0, H, W), 1).requires_grad_(True)
    x1 = seeded((B, C1, H, W), 2).requires_grad_(True) if C1 else None
    w = (seeded((Cout, C0 + C1, k, k), 3) * (1.0 / (k * (C0 + C1) ** 0.5))).requires_grad_(True)
    b = seeded((Cout,), 4, 0.1).requires_grad_(True)
    x = x0 if x1 is None else torch.cat((x0, x1), dim=1)
    if up2:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, w, b, padding=pad)
    dy = seeded(tuple(y.shape), 5)
    y.backward(dy)
    lib = _lib.load()
    d0, d1, dw, db = torch.empty((B, C0, H, W), device=DEV), (torch.empty((B, C1, H, W), device=DEV) if C1 else None), \
        torch.empty(w.shape, device=DEV), torch.empty((Cout,), device=DEV)
    a = [dev(t.detach()) for t in (x0, w)] + [dev(x1.detach()) if C1 else None, dev(dy)]
    _lib.check(lib.dm_op_conv2d_bwd(_lib.ptr(a[0]), C0, _lib.ptr(a[2]), C1, _lib.ptr(a[1]), _lib.ptr(a[3]), _lib.ptr(d0),
                                    _lib.ptr(d1), _lib.ptr(dw), _lib.ptr(db), B, H, W, Cout, k, pad, int(up2), None))
    errs = dict(dx0=rel_l2(d0.cpu(), x0.grad), dw=rel_l2(dw.cpu(), w.grad), db=rel_l2(db.cpu(), b.grad))
    if C1:
        errs["dx1"] = rel_l2(d1.cpu(), x1.grad)
    print(case, errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("case", [(2, 64, 32, 32, 64), (3, 64, 16, 16, 128), (2, 32, 8, 12, 48), (1, 128, 4, 4, 256)])
def test_downsample_bwd(case):
    B, C, H, W, Cout = case
    x = seeded((B, C, H, W), 1).requires_grad_(True)
    sd = {"d.1.weight": (seeded((Cout, 4 * C, 1, 1), 2) * 0.05).requires_grad_(True),
          "d.1.bias": seeded((Cout,), 3, 0.1).requires_grad_(True)}
    y = uo.downsample(sd, "d", x)
    dy = seeded(tuple(y.shape), 4)
    y.backward(dy)
    lib = _lib.load()
    dx, dw, db = torch.empty(x.shape, device=DEV), torch.empty((Cout, 4 * C, 1, 1), device=DEV), torch.empty((Cout,), device=DEV)
    a = [dev(x.detach()), dev(sd["d.1.weight"].detach()), dev(dy)]
    _lib.check(lib.dm_op_downsample_bwd(_lib.ptr(a[0]), C, _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(dx), _lib.ptr(dw),
                                        _lib.ptr(db), B, H, W, Cout, None))
    errs = dict(dx=rel_l2(dx.cpu(), x.grad), dw=rel_l2(dw.cpu(), sd["d.1.weight"].grad), db=rel_l2(db.cpu(), sd["d.1.bias"].grad))
    print(case, errs)
    assert max(errs.values()) < TOL, errs


BLOCK_BWD_CASES = [
    # (B, Cin, Cout, H, W, scale_shift)
    (2, 64, 64, 32, 32, True), (3, 128, 64, 16, 16, False), (2, 192, 128, 8, 8, True), (4, 512, 512, 4, 4, True),
    (2, 32, 32, 16, 16, True), (2, 24, 40, 7, 5, True), (1, 768, 512, 4, 4, False), (2, 64, 1024, 2, 2, True),
]


@pytest.mark.parametrize("case", BLOCK_BWD_CASES)
def test_block_bwd(case):
    B, Cin, Cout, H, W, ss = case
    x = seeded((B, Cin, H, W), 1).requires_grad_(True)
    sd = {"b.proj.weight": (seeded((Cout, Cin, 3, 3), 2) / (3 * Cin ** 0.5)).requires_grad_(True),
          "b.proj.bias": seeded((Cout,), 3, 0.1).requires_grad_(True),
          "b.norm.g": (1 + 0.25 * seeded((1, Cout, 1, 1), 4)).requires_grad_(True)}
    scale = seeded((B, Cout, 1, 1), 5, 0.5).requires_grad_(True) if ss else None
    shift = seeded((B, Cout, 1, 1), 6, 0.5).requires_grad_(True) if ss else None
    y = uo.block(sd, "b", x, (scale, shift) if ss else None)
    dy = seeded(tuple(y.shape), 7)
    y.backward(dy)
    lib = _lib.load()
    outs = dict(dx=torch.empty(x.shape, device=DEV), dw=torch.empty((Cout, Cin, 3, 3), device=DEV),
                db=torch.empty((Cout,), device=DEV), dg=torch.empty((Cout,), device=DEV),
                dsc=torch.empty((B, Cout), device=DEV) if ss else None, dsh=torch.empty((B, Cout), device=DEV) if ss else None)
    a = [dev(x.detach()), dev(sd["b.proj.weight"].detach()), dev(sd["b.proj.bias"].detach()), dev(sd["b.norm.g"].detach()),
         dev(scale.detach().reshape(B, Cout)) if ss else None, dev(shift.detach().reshape(B, Cout)) if ss else None, dev(dy)]
    _lib.check(lib.dm_op_block_bwd(_lib.ptr(a[0]), Cin, _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]), _lib.ptr(a[4]),
                                   _lib.ptr(a[5]), _lib.ptr(a[6]), _lib.ptr(outs["dx"]), _lib.ptr(outs["dw"]),
                                   _lib.ptr(outs["db"]), _lib.ptr(outs["dg"]), _lib.ptr(outs["dsc"]), _lib.ptr(outs["dsh"]),
                                   B, H, W, Cout, None))
    errs = dict(dx=rel_l2(outs["dx"].cpu(), x.grad), dw=rel_l2(outs["dw"].cpu(), sd["b.proj.weight"].grad),
                db=rel_l2(outs["db"].cpu(), sd["b.proj.bias"].grad), dg=rel_l2(outs["dg"].cpu(), sd["b.norm.g"].grad.reshape(-1)))
    if ss:
        errs["dscale"] = rel_l2(outs["dsc"].cpu(), scale.grad.reshape(B, Cout))
        errs["dshift"] = rel_l2(outs["dsh"].cpu(), shift.grad.reshape(B, Cout))
    print(case, errs)
    assert max(errs.values()) < TOL, errs


def test_rmsnorm_bwd_incl_zero_rows():
    """F.normalize clamps the norm at 1e-12: an all-zero pixel has gradient dy * g * sqrt(C) / 1e-12, no projection term."""
    lib = _lib.load()
    for (B, C, H, W) in ((2, 64, 8, 8), (3, 36, 5, 7), (1, 1024, 2, 2)):
        x = seeded((B, C, H, W), 1)
        x[0, :, 0, 0] = 0.0
        x.requires_grad_(True)
        g = (1 + 0.25 * seeded((1, C, 1, 1), 2)).requires_grad_(True)
        # the reference's expression itself (DD/denoising_diffusion.py:67): F.normalize's backward is finite at a zero
        # vector, the oracle's x / sqrt(sum x^2).clamp_min(eps) restatement differentiates to NaN there
        y = F.normalize(x, dim=1) * g * (C ** 0.5)
        dy = seeded(tuple(y.shape), 3) * 1e-6  # keeps the zero row's gradient finite in fp32
        y.backward(dy)
        dx, dg = torch.empty(x.shape, device=DEV), torch.empty((C,), device=DEV)
        a = [dev(x.detach()), dev(g.detach()), dev(dy)]
        _lib.check(lib.dm_op_rmsnorm_bwd(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(dx), _lib.ptr(dg), B, C, H, W,
                                         None))
        assert rel_l2(dx.cpu(), x.grad) < TOL and rel_l2(dg.cpu(), g.grad.reshape(-1)) < TOL


def _attn_params(C, full, seed):
    hid = 128
    sd = {"a.norm.g": 1 + 0.25 * seeded((1, C, 1, 1), seed),
          "a.mem_kv": seeded((2, 4, 4, 32) if full else (2, 4, 32, 4), seed + 1),
          "a.to_qkv.weight": seeded((3 * hid, C, 1, 1), seed + 2) / C ** 0.5}
    if full:
        sd["a.to_out.weight"] = seeded((C, hid, 1, 1), seed + 3) / hid ** 0.5
        sd["a.to_out.bias"] = seeded((C,), seed + 4, 0.1)
    else:
        sd["a.to_out.0.weight"] = seeded((C, hid, 1, 1), seed + 3) / hid ** 0.5
        sd["a.to_out.0.bias"] = seeded((C,), seed + 4, 0.1)
        sd["a.to_out.1.g"] = 1 + 0.25 * seeded((1, C, 1, 1), seed + 5)
    return {k: v.requires_grad_(True) for k, v in sd.items()}


@pytest.mark.parametrize("case", [(2, 64, 32, 32), (2, 64, 16, 16), (3, 128, 8, 8), (2, 256, 8, 8), (2, 32, 5, 7), (1, 64, 64, 64)])
def test_linear_attention_bwd(case):
    B, C, H, W = case
    sd = _attn_params(C, False, 10)
    x = seeded((B, C, H, W), 1).requires_grad_(True)
    y = uo.linear_attention(sd, "a", x, 4, 32)
    dy = seeded(tuple(y.shape), 2)
    y.backward(dy)
    lib = _lib.load()
    names = ["a.norm.g", "a.mem_kv", "a.to_qkv.weight", "a.to_out.0.weight", "a.to_out.0.bias", "a.to_out.1.g"]
    ins = [dev(x.detach())] + [dev(sd[k].detach()) for k in names] + [dev(dy)]
    dx = torch.empty(x.shape, device=DEV)
    outs = [torch.empty(sd[k].shape, device=DEV) for k in names]
    _lib.check(lib.dm_op_linear_attention_bwd(*[_lib.ptr(t) for t in ins], _lib.ptr(dx), *[_lib.ptr(t) for t in outs],
                                              B, C, H, W, 4, 32, None))
    errs = {"dx": rel_l2(dx.cpu(), x.grad)}
    errs.update({k: rel_l2(o.cpu(), sd[k].grad) for k, o in zip(names, outs)})
    print(case, errs)
    assert max(errs.values()) < TOL, errs


@pytest.mark.parametrize("case", [(2, 256, 4, 4), (3, 512, 4, 4), (2, 512, 8, 8), (2, 64, 3, 5), (1, 128, 16, 16), (2, 64, 1, 1),
                                  (1, 64, 32, 16), (2, 32, 23, 25)])  # the last two: the tiled form (over ~300 tokens)
def test_attention_bwd(case):
    B, C, H, W = case
    sd = _attn_params(C, True, 20)
    x = seeded((B, C, H, W), 1).requires_grad_(True)
    y = uo.full_attention(sd, "a", x, 4, 32)
    dy = seeded(tuple(y.shape), 2)
    y.backward(dy)
    lib = _lib.load()
    names = ["a.norm.g", "a.mem_kv", "a.to_qkv.weight", "a.to_out.weight", "a.to_out.bias"]
    ins = [dev(x.detach())] + [dev(sd[k].detach()) for k in names] + [dev(dy)]
    dx = torch.empty(x.shape, device=DEV)
    outs = [torch.empty(sd[k].shape, device=DEV) for k in names]
    _lib.check(lib.dm_op_attention_bwd(*[_lib.ptr(t) for t in ins], _lib.ptr(dx), *[_lib.ptr(t) for t in outs],
                                       B, C, H, W, 4, 32, None))
    errs = {"dx": rel_l2(dx.cpu(), x.grad)}
    errs.update({k: rel_l2(o.cpu(), sd[k].grad) for k, o in zip(names, outs)})
    print(case, errs)
    assert max(errs.values()) < TOL, errs


def test_conv_backward_random_shapes():
    """Seeded random shapes through the weight-gradient kernel's tiling (ragged pixel blocks, several cin / cout tiles,
    K splits) and through whichever forward kernel the input-gradient convolution dispatches to."""
    import random

    import os

    rng = random.Random(int(os.environ.get("DM_TEST_SEED", "2024")))
    lib = _lib.load()
    # shapes a longer sweep found: 3x3 on a 1x1 map with channel counts off the Winograd kernels' grid (the direct kernel's
    # window of a 256-pixel tile is nine times the tile: narrower pixel tiles)
    fixed = [(3, 2, 1, 1, 44, 0, 192, False), (3, 5, 2, 2, 20, 12, 100, False), (3, 17, 1, 1, 36, 0, 40, True)]
    n_random = int(os.environ.get("DM_TEST_SHAPES", "20"))  # DM_TEST_SEED / DM_TEST_SHAPES: longer off-line sweeps
    for it in range(n_random + len(fixed)):
        k = rng.choice([3, 3, 1])
        B = rng.choice([1, 2, 3, 5, 17])
        H, W = rng.choice([(4, 4), (8, 8), (6, 10), (16, 16), (12, 20), (32, 32), (3, 3), (1, 1), (2, 2), (2, 6), (64, 64),
                           (10, 34), (14, 14)])
        C0 = 4 * rng.randint(1, 40)
        C1 = rng.choice([0, 0, 4 * rng.randint(1, 24)])
        Cout = 4 * rng.randint(1, 48)
        up2 = k == 3 and C1 == 0 and rng.random() < 0.25
        if it >= n_random:
            k, B, H, W, C0, C1, Cout, up2 = fixed[it - n_random]
        x0 = seeded((B, C0, H, W), 100 + it).requires_grad_(True)
        x1 = seeded((B, C1, H, W), 200 + it).requires_grad_(True) if C1 else None
        w = (seeded((Cout, C0 + C1, k, k), 300 + it) / (k * (C0 + C1) ** 0.5)).requires_grad_(True)
        b = seeded((Cout,), 400 + it, 0.1).requires_grad_(True)
        x = x0 if x1 is None else torch.cat((x0, x1), dim=1)
        if up2:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        y = F.conv2d(x, w, b, padding=k // 2)
        dy = seeded(tuple(y.shape), 500 + it)
        y.backward(dy)
        d0 = torch.empty((B, C0, H, W), device=DEV)
        d1 = torch.empty((B, C1, H, W), device=DEV) if C1 else None
        dw, db = torch.empty(w.shape, device=DEV), torch.empty((Cout,), device=DEV)
        a = [dev(x0.detach()), dev(w.detach()), dev(x1.detach()) if C1 else None, dev(dy)]
        _lib.check(lib.dm_op_conv2d_bwd(_lib.ptr(a[0]), C0, _lib.ptr(a[2]), C1, _lib.ptr(a[1]), _lib.ptr(a[3]), _lib.ptr(d0),
                                        _lib.ptr(d1), _lib.ptr(dw), _lib.ptr(db), B, H, W, Cout, k, k // 2, int(up2), None))
        errs = dict(dx0=rel_l2(d0.cpu(), x0.grad), dw=rel_l2(dw.cpu(), w.grad), db=rel_l2(db.cpu(), b.grad))
        if C1:
            errs["dx1"] = rel_l2(d1.cpu(), x1.grad)
        assert max(errs.values()) < TOL, ((B, C0, C1, H, W, Cout, k, up2), errs)


@pytest.mark.parametrize("R,I,O", [(64, 256, 8064), (16, 64, 256), (20, 256, 256), (100, 512, 64), (7, 256, 128), (64, 48, 80)])
def test_linear_forward_and_backward_vs_torch(R, I, O):
    """The batch-row nn.Linear layers of the training step (time_mlp, the concatenated ResnetBlock.mlp matrix) through
    dm_op_linear / dm_op_linear_bwd: MFMA GEMMs (small_gemm.hip) where the shapes allow -- 16-byte rows, I % 16 / 64, O % 16 / 64,
    ragged row counts -- and the VALU kernels otherwise ((7, ...) rows, (.., 48, 80)); against torch on the CPU in double."""
    import ctypes as C

    from diffusion_models_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(R * 1000 + I + O)
    x = torch.randn((R, I), generator=g)
    w = torch.randn((O, I), generator=g) / I ** 0.5
    b = torch.randn((O,), generator=g)
    dy = torch.randn((R, O), generator=g)
    xd, wd, bd, dyd = (t.to(DEV) for t in (x, w, b, dy))
    y = torch.empty((R, O), device=DEV)
    _lib.check(lib.dm_op_linear(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), R, I, O, None))
    want = x.double() @ w.double().T + b.double()
    assert rel_l2(y.cpu(), want.float()) < 2e-6
    dx = torch.empty((R, I), device=DEV)
    dw = torch.full((O, I), 0.5, device=DEV)
    db = torch.full((O,), -0.25, device=DEV)
    for acc in (0, 1):  # overwrite, then accumulate on top of the first result
        _lib.check(lib.dm_op_linear_bwd(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(dyd), _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db),
                                        R, I, O, acc, None))
        k = acc + 1
        assert rel_l2(dx.cpu(), (dy.double() @ w.double()).float()) < 2e-6
        assert rel_l2(dw.cpu(), (k * (dy.double().T @ x.double())).float()) < 2e-6
        assert rel_l2(db.cpu(), (k * dy.double().sum(0)).float()) < 2e-6
