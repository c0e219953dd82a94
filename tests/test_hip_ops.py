"""GPU parity of the single HIP operators (called through the C ABI, dm_op_*) against the
oracle / plain PyTorch fp32 on the CPU, on the same seeded inputs.

Tolerance: fp32 with a different summation order -> rel-L2 <= 2e-5 per operator
(measured ~1e-6); the end-to-end budget of the task is 1e-3."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

import diffusion_models_amd as dm
from diffusion_models_amd import _lib
from oracle import unet_oracle as uo

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-5
DEV = "cuda:0"


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def dev(t):
    return None if t is None else t.to(DEV).contiguous()


def hip_conv(x0, w, b=None, x1=None, residual=None, pad=0, up2=False):
    lib = _lib.load()
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    Cout, _, k, _ = w.shape
    Hin, Win = (2 * H, 2 * W) if up2 else (H, W)
    Ho, Wo = Hin + 2 * pad - k + 1, Win + 2 * pad - k + 1
    out = torch.empty((B, Cout, Ho, Wo), device=DEV)
    args = [dev(t) for t in (x0, x1, w, b, residual)]
    _lib.check(lib.dm_op_conv2d(_lib.ptr(args[0]), C0, _lib.ptr(args[1]), C1, _lib.ptr(args[2]), _lib.ptr(args[3]),
                                _lib.ptr(args[4]), _lib.ptr(out), B, H, W, Cout, k, pad, int(up2), None))
    return out.cpu()


CONV_CASES = [
    # (B, C0, C1, H, W, Cout, k, pad, up2, bias, residual)
    (2, 64, 0, 32, 32, 64, 3, 1, False, True, False),     # Block conv @32^2, tile 8x32
    (3, 64, 64, 16, 16, 64, 3, 1, False, True, True),     # concat + residual, whole-image tile
    (2, 128, 64, 8, 8, 128, 3, 1, False, True, False),    # 2x2 wave grid, NB=2 images per tile
    (5, 256, 0, 4, 4, 256, 3, 1, False, True, False),     # 1x4 wave grid, NB=4, ragged batch
    (2, 256, 0, 4, 4, 512, 3, 1, False, True, False),     # two N tiles
    (1, 512, 256, 4, 4, 512, 3, 1, False, True, False),   # deepest layer: K = 6912
    (2, 32, 0, 8, 8, 16, 3, 1, True, True, False),        # nearest x2 folded into the gather
    (2, 64, 0, 16, 16, 384, 1, 0, False, False, False),   # to_qkv 1x1, three N tiles, no bias
    (2, 128, 0, 8, 8, 64, 1, 0, False, True, True),       # to_out 1x1 + residual
    (2, 3, 0, 32, 32, 64, 7, 3, False, True, False),      # init_conv: thin input, CK=4 path
    (2, 4, 0, 16, 16, 32, 7, 3, False, True, False),      # latent init_conv
    (3, 4, 0, 20, 28, 64, 7, 3, False, True, False),      # first-conv kernel (init7_mfma.hip): ragged 16x16 blocks, 4 channels
    (1, 6, 0, 8, 8, 64, 7, 3, False, False, False),       # image smaller than a block, 6 channels, no bias
    (2, 8, 0, 16, 16, 64, 7, 3, False, True, False),      # 8 channels: 98 K steps
    (2, 64, 0, 32, 32, 3, 1, 0, False, True, False),      # final_conv: 3 output channels
    (1, 64, 0, 64, 64, 64, 3, 1, False, True, False),     # 64x64 image: 2 tiles across, 8 down
    (2, 16, 0, 24, 24, 16, 3, 1, False, True, False),     # non power-of-two image (masked tile)
    (2, 48, 16, 12, 12, 48, 3, 1, False, True, False),    # odd sizes, concat
    (1, 6, 0, 8, 8, 8, 3, 1, False, True, False),         # channels not a multiple of 4
    # Winograd F(2x2,3x3) path (3x3, C % 8 == 0, Cout % 64 == 0, even size); the cases above with those
    # properties take it too
    (3, 72, 8, 12, 20, 64, 3, 1, False, True, True),      # non power-of-two even image, masked tiles, concat 72+8
    (37, 64, 0, 4, 4, 128, 3, 1, False, False, False),    # 8 images per workgroup, ragged batch, no bias
    (2, 8, 0, 6, 2, 64, 3, 1, False, True, False),        # single K chunk, image narrower than a tile row
    (2, 1024, 0, 4, 4, 64, 3, 1, False, True, False),     # Winograd with K split 8 ways (partial sums + finalize)
    (1, 1024, 0, 2, 2, 64, 3, 1, False, True, False),     # 2x2 image: window too large for Winograd -> direct kernel
    (300, 192, 0, 1, 1, 44, 3, 1, False, True, False),    # 3x3 on 1x1 maps off the Winograd grid: the direct kernel's window is 9x its pixel tile (narrower tiles)
    (70, 40, 24, 2, 2, 100, 3, 1, False, True, True),     # ... 2x2 maps, concat, residual
    # nearest x2 + 3x3 on the source grid (upwino_mfma.hip; C % 8 == 0, Cout % 64 == 0, source 4x4 or multiples of 8).
    # Small shapes reach it only with DM_UPWINO_MIN_WGS=1 DM_UPWINO_MIN_K=1 (tests/test_hip_forced_dispatch.py),
    # otherwise they check the folded direct kernel
    (2, 64, 0, 8, 8, 64, 3, 1, True, True, False),        # one 8x8 block per image
    (3, 128, 0, 16, 16, 128, 3, 1, True, True, True),     # 2x2 blocks, two cout tiles, residual
    (5, 64, 0, 4, 4, 128, 3, 1, True, True, False),       # four 4x4 images per workgroup, ragged batch
    (2, 512, 0, 4, 4, 64, 3, 1, True, True, False),       # K split (partial sums + finalize)
    (1, 16, 0, 8, 24, 64, 3, 1, True, False, False),      # non-square source, two chunks, no bias
    (40, 128, 0, 16, 16, 64, 3, 1, True, True, False),    # 160 workgroups of 16 chunks: production dispatch
    # 1x1 GEMM kernel (pw_mfma.hip; C % 8 == 0, Cout % 64 == 0): res_conv / attention projections
    (3, 64, 64, 10, 6, 64, 1, 0, False, True, True),      # two sources + residual, 180 pixels (ragged last rows)
    (2, 512, 256, 4, 4, 512, 1, 0, False, True, True),    # res_conv of the deepest stage: 96 chunks, K split, 4 cout blocks
    (5, 256, 0, 8, 8, 384, 1, 0, False, False, False),    # to_qkv: three 128-cout blocks, no bias
    (1, 8, 0, 2, 2, 64, 1, 0, False, True, False),        # one chunk, four pixels
    (7, 72, 8, 16, 16, 128, 1, 0, False, True, False),    # chunk boundary of the concat inside the prefetch ring
    (2, 64, 0, 8, 8, 192, 1, 0, False, True, False),      # 192 couts: three 64-cout blocks (InceptionV3 branches)
    # even images >= 16x16 with C % 16 == 0, Cout % 64 == 0: masked / ragged pixel blocks of the Winograd kernels
    (2, 32, 16, 24, 40, 128, 3, 1, False, True, True),    # masked 16x16 blocks, concat 32+16, two cout tiles, residual
    (1, 16, 0, 18, 34, 64, 3, 1, False, False, False),    # single chunk, no bias, ragged blocks in both directions
    # the direct 3x3 kernel behind it (odd sizes are not Winograd-eligible)
    (2, 64, 64, 7, 9, 64, 3, 1, False, True, True),
    (3, 256, 0, 5, 5, 256, 3, 1, False, True, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d(case):
    B, C0, C1, H, W, Cout, k, pad, up2, bias, residual = case
    x0 = seeded((B, C0, H, W), 1)
    x1 = seeded((B, C1, H, W), 2) if C1 else None
    w = seeded((Cout, C0 + C1, k, k), 3, (C0 + C1) ** -0.5 / k)
    b = seeded((Cout,), 4) if bias else None
    xin = x0 if x1 is None else torch.cat((x0, x1), 1)
    if up2:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = F.conv2d(xin, w, b, padding=pad)
    res = seeded(ref.shape, 5) if residual else None
    if residual:
        ref = ref + res
    got = hip_conv(x0, w, b, x1, res, pad, up2)
    assert got.shape == ref.shape
    assert rel_l2(got, ref) < TOL


@pytest.mark.parametrize("shape", [(2, 32, 16, 16, 64), (3, 64, 8, 8, 128), (2, 128, 4, 4, 256), (1, 16, 6, 10, 16)])
def test_downsample(shape):
    B, Cc, H, W, Cout = shape
    x = seeded((B, Cc, H, W), 1)
    sd = {"d.1.weight": seeded((Cout, 4 * Cc, 1, 1), 2, (4 * Cc) ** -0.5), "d.1.bias": seeded((Cout,), 3)}
    ref = uo.downsample(sd, "d", x)
    out = torch.empty(ref.shape, device=DEV)
    a = [dev(x), dev(sd["d.1.weight"]), dev(sd["d.1.bias"])]
    _lib.check(_lib.load().dm_op_downsample(_lib.ptr(a[0]), Cc, _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(out), B, H, W,
                                            Cout, None))
    assert rel_l2(out.cpu(), ref) < TOL


@pytest.mark.parametrize("shape", [(2, 64, 8, 8), (1, 512, 4, 4), (3, 48, 5, 7), (2, 3, 4, 4)])
def test_rmsnorm(shape):
    B, Cc, H, W = shape
    x = seeded(shape, 1)
    g = 1 + 0.3 * seeded((1, Cc, 1, 1), 2)
    ref = uo.rms_norm(x, g)
    out = torch.empty(shape, device=DEV)
    a = [dev(x), dev(g)]
    _lib.check(_lib.load().dm_op_rmsnorm(_lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(out), B, Cc, H, W, None))
    assert rel_l2(out.cpu(), ref) < TOL
    # zero row: F.normalize clamps the norm at 1e-12 -> output 0, not NaN
    x[0, :, 0, 0] = 0
    ax = dev(x)
    _lib.check(_lib.load().dm_op_rmsnorm(_lib.ptr(ax), _lib.ptr(a[1]), _lib.ptr(out), B, Cc, H, W, None))
    assert torch.isfinite(out).all() and rel_l2(out.cpu(), uo.rms_norm(x, g)) < TOL


BLOCK_CASES = [
    # (B, Cin, Cout, H, W, scale_shift)
    (2, 64, 64, 32, 32, True),     # fused epilogue, one wave column
    (2, 64, 64, 16, 16, False),
    (3, 192, 128, 16, 16, True),   # fused, cross-wave reduction over 2 waves
    (2, 384, 256, 8, 8, True),     # fused, cross-wave reduction over 4 waves
    (5, 256, 256, 4, 4, True),     # NB=4 images per tile, per-image scale/shift, ragged
    (2, 768, 512, 4, 4, True),     # Cout > 256: conv + separate norm kernel
    (2, 32, 48, 8, 8, True),       # Cout not a multiple of 32
    (5, 64, 64, 4, 4, True),       # Winograd, 8 images per workgroup: per-image scale/shift rows inside one tile
    (3, 64, 64, 12, 10, True),     # Winograd fused epilogue on masked tiles
    (2, 64, 64, 7, 9, True),       # direct kernel, fused epilogue (odd size)
    (2, 128, 128, 5, 5, False),    # direct kernel, 2x2 wave grid
    (2, 192, 256, 3, 3, True),     # direct kernel, 1x4 wave grid
]


@pytest.mark.parametrize("case", BLOCK_CASES, ids=[str(c) for c in BLOCK_CASES])
def test_block(case):
    B, Cin, Cout, H, W, ss = case
    x = seeded((B, Cin, H, W), 1)
    sd = {
        "b.proj.weight": seeded((Cout, Cin, 3, 3), 2, (9 * Cin) ** -0.5),
        "b.proj.bias": seeded((Cout,), 3, 0.1),
        "b.norm.g": 1 + 0.3 * seeded((1, Cout, 1, 1), 4),
    }
    scale = seeded((B, Cout), 5, 0.5) if ss else None
    shift = seeded((B, Cout), 6, 0.5) if ss else None
    ref = uo.block(sd, "b", x, (scale[:, :, None, None], shift[:, :, None, None]) if ss else None)
    out = torch.empty(ref.shape, device=DEV)
    a = [dev(t) for t in (x, sd["b.proj.weight"], sd["b.proj.bias"], sd["b.norm.g"], scale, shift)]
    _lib.check(_lib.load().dm_op_block(_lib.ptr(a[0]), Cin, _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]),
                                       _lib.ptr(a[4]), _lib.ptr(a[5]), _lib.ptr(out), B, H, W, Cout, None))
    assert rel_l2(out.cpu(), ref) < TOL


def _attn_sd(Cc, full, heads=4, dh=32, seed=10):
    hid = heads * dh
    sd = {
        "a.norm.g": 1 + 0.3 * seeded((1, Cc, 1, 1), seed),
        "a.mem_kv": seeded((2, heads, 4, dh) if full else (2, heads, dh, 4), seed + 1),
        "a.to_qkv.weight": seeded((3 * hid, Cc, 1, 1), seed + 2, Cc ** -0.5),
    }
    if full:
        sd["a.to_out.weight"] = seeded((Cc, hid, 1, 1), seed + 3, hid ** -0.5)
        sd["a.to_out.bias"] = seeded((Cc,), seed + 4, 0.1)
    else:
        sd["a.to_out.0.weight"] = seeded((Cc, hid, 1, 1), seed + 3, hid ** -0.5)
        sd["a.to_out.0.bias"] = seeded((Cc,), seed + 4, 0.1)
        sd["a.to_out.1.g"] = 1 + 0.3 * seeded((1, Cc, 1, 1), seed + 5)
    return sd


# C in {64, 128} take the two fused kernels (linattn_fused.hip), the others the unfused chain of kernels
@pytest.mark.parametrize("shape", [(2, 64, 32, 32), (2, 64, 16, 16), (3, 128, 8, 8), (1, 256, 8, 8), (2, 32, 6, 10),
                                   (2, 64, 6, 10), (3, 128, 16, 16), (1, 64, 2, 2), (2, 128, 18, 14)])
def test_linear_attention(shape):
    B, Cc, H, W = shape
    x = seeded(shape, 1)
    sd = _attn_sd(Cc, full=False)
    ref = uo.linear_attention(sd, "a", x, 4, 32)
    out = torch.empty(shape, device=DEV)
    a = [dev(t) for t in (x, sd["a.norm.g"], sd["a.mem_kv"], sd["a.to_qkv.weight"], sd["a.to_out.0.weight"],
                          sd["a.to_out.0.bias"], sd["a.to_out.1.g"])]
    _lib.check(_lib.load().dm_op_linear_attention(*[_lib.ptr(t) for t in a], _lib.ptr(out), B, Cc, H, W, 4, 32, None))
    assert rel_l2(out.cpu(), ref) < TOL


# 4x4 maps with C % 256 == 0 take the one-kernel path (attn16_fused.hip), the others the unfused chain
@pytest.mark.parametrize("shape", [(2, 256, 4, 4), (3, 512, 4, 4), (70, 512, 4, 4), (1, 1024, 4, 4), (1, 512, 8, 8),
                                   (2, 64, 2, 2), (2, 128, 4, 4), (1, 128, 16, 16),
                                   (1, 64, 32, 32), (2, 32, 25, 27)])  # over ~590 tokens: the tiled attention core
def test_attention(shape):
    B, Cc, H, W = shape
    x = seeded(shape, 1)
    sd = _attn_sd(Cc, full=True)
    ref = uo.full_attention(sd, "a", x, 4, 32)
    out = torch.empty(shape, device=DEV)
    a = [dev(t) for t in (x, sd["a.norm.g"], sd["a.mem_kv"], sd["a.to_qkv.weight"], sd["a.to_out.weight"],
                          sd["a.to_out.bias"])]
    _lib.check(_lib.load().dm_op_attention(*[_lib.ptr(t) for t in a], _lib.ptr(out), B, Cc, H, W, 4, 32, None))
    assert rel_l2(out.cpu(), ref) < TOL


def test_sampler_update_bit_exact():
    """The update is elementwise fp32 with contraction off: it must equal the reference's
    expression tree bit for bit (clamp included), for every objective (pred_noise / pred_x0 / pred_v,
    DD/denoising_diffusion.py:607-624), and hand back the clamped x_start."""
    n = 3 * 32 * 32 * 2 + 3  # not a multiple of 4
    x, eps, z = seeded((n,), 1), seeded((n,), 2, 3.0), seeded((n,), 3)
    lib = _lib.load()
    for obj in (0, 1, 2):
        for kind, c in ((0, [1.7, 1.3, 0.4, 0.6, 0.2, 1.0, 0.59, 0.81]), (0, [1.0, 0.01, 1.0, 0.0, 1e-10, 0.0, 0.99, 0.1]),
                        (1, [3.5, 3.4, 0.9, 0.43, 0.1, 1.0, 0.28, 0.96]), (1, [1.2, 0.6, 0, 0, 0, 0.0, 0.83, 0.55])):
            ct = torch.tensor(c, dtype=torch.float32)
            raw = (ct[0] * x - ct[1] * eps, eps, ct[6] * x - ct[7] * eps)[obj]
            x0 = raw.clamp(-1.0, 1.0)
            if kind == 0:
                mean = ct[2] * x0 + ct[3] * x
                ref = mean + ct[4] * z if c[5] else mean + ct[4] * 0.0
            else:
                e2 = (ct[0] * x - x0) / ct[1]
                ref = (x0 * ct[2] + ct[3] * e2) + ct[4] * z if c[5] else x0
            out = torch.empty(n, device=DEV)
            xs = torch.empty(n, device=DEV)
            a = [dev(x), dev(eps), dev(z)]
            carr = (C.c_float * 8)(*c)
            _lib.check(lib.dm_op_sampler_update(kind, obj, _lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), carr,
                                                _lib.ptr(out), _lib.ptr(xs), n, None))
            assert torch.equal(out.cpu(), ref), (obj, kind, c)
            assert torch.equal(xs.cpu(), x0), (obj, kind, c)


def test_philox_noise_statistics_and_determinism():
    lib = _lib.load()
    n = 1 << 22
    a = torch.empty(n, device=DEV)
    b = torch.empty(n, device=DEV)
    _lib.check(lib.dm_randn(_lib.ptr(a), n, 1234, 0, 0, None))
    _lib.check(lib.dm_randn(_lib.ptr(b), n, 1234, 0, 0, None))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # element offset: the tail of the stream generated on its own equals the tail of the whole stream
    off = 12 * 1024 + 4
    _lib.check(lib.dm_randn(_lib.ptr(b), n - off, 1234, 0, off, None))
    torch.cuda.synchronize()
    assert torch.equal(b[: n - off], a[off:])
    with pytest.raises(RuntimeError):
        _lib.check(lib.dm_randn(_lib.ptr(b), 16, 1234, 0, 3, None))  # offsets are multiples of 4
    _lib.check(lib.dm_randn(_lib.ptr(b), n, 1234, 1, 0, None))
    torch.cuda.synchronize()
    a, b = a.cpu().double(), b.cpu().double()
    assert abs(a.mean()) < 3e-3 and abs(a.std() - 1) < 3e-3
    assert abs((a ** 3).mean()) < 1e-2 and abs((a ** 4).mean() - 3) < 3e-2
    assert abs((a * b).mean()) < 3e-3  # different draws are uncorrelated
    assert torch.isfinite(a).all()
