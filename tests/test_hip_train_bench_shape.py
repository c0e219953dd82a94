"""GPU parity of the training step AT THE SHAPES bench.py TIMES (``train_step``: the full 35.7 M-parameter U-Net, 32x32,
batch 64, dropout 0.1 -- ddpm_cifar.yaml) and at two more batches whose tiling plans differ: the grouped weight-gradient
split table, the pixel-block sizes, ``conv_plan`` / ``wino_plan`` and the deferred reductions are all functions of B*H*W, so
a B=8 fixture does not exercise what a B=64 run takes (tests/test_hip_configs.py makes the same point for sampling).

Oracle: ``oracle/train_oracle.py`` (torch autograd through the CPU restatement, pinned to the reference's own autograd by
tests/test_oracle_golden.py).  Tolerances as in test_hip_train.py: loss 1e-5 relative, every gradient tensor 2e-4 rel-L2."""
import ctypes as C

import pytest
import torch

import diffusion_models_amd as dm
from diffusion_models_amd import _lib
from diffusion_models_amd.spec import UnetConfig

from conftest import rel_l2
from test_hip_train import _block_shapes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_TOL = 2e-4
CFG = UnetConfig()


def _data(B, seed):
    g = torch.Generator().manual_seed(seed)
    x_start = torch.rand((B, 3, 32, 32), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 3, 32, 32), generator=g)
    return x_start, t, noise


def _model(sd, dropout=0.0):
    u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, dropout=dropout, device=DEV)
    u.load_state_dict(sd)
    return u, dm.DenoisingDiffusion(u, image_size=32, timesteps=1000).train()


def _compare(grads, want, what):
    assert set(grads) == set(want)
    worst = max((rel_l2(grads[k].cpu(), want[k]), k) for k in want)
    print(what, "worst gradient", worst)
    assert worst[0] < GRAD_TOL, worst


def test_bench_workload_b64_dropout_vs_oracle():
    """Loss and all 245 gradients of the workload ``bench.py`` times: B=64, dropout 0.1, the library's Philox masks exported
    (dm_op_dropout_mask) into the oracle's Blocks."""
    from oracle import train_oracle as to

    sd = dm.synth_state_dict(dm.unet_param_spec(CFG), salt=0)
    u, d = _model(sd, dropout=0.1)
    p, seed, B = 0.1, 20261005, 64
    u.set_dropout_seed(seed)
    x_start, t, noise = _data(B, 64)
    loss = float(d.p_losses(x_start, t, noise=noise))
    grads = d.model.grads()
    lib = _lib.load()
    masks = []
    for k, (c, h, w) in enumerate(_block_shapes(CFG, 32)):
        m = torch.empty((B, h, w, c), device=DEV)
        _lib.check(lib.dm_op_dropout_mask(_lib.ptr(m), m.numel(), p, C.c_uint64(seed), C.c_uint64(0), k, None))
        masks.append(m.permute(0, 3, 1, 2).contiguous().cpu())
    torch.set_num_threads(16)
    want_loss, want = to.loss_and_grads(sd, CFG, dm.make_schedule(1000, "linear"), x_start, t, noise, dropout_masks=masks)
    print("B=64 dropout 0.1: loss", loss, want_loss)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    _compare(grads, want, "B=64 dropout 0.1")


@pytest.mark.parametrize("B", [16, 128])
def test_other_batches_vs_oracle(B):
    """The same network without dropout at B=16 and B=128: other K-split counts, weight-gradient split tables and norm
    chunkings than the B=8 fixture and the B=64 workload."""
    from oracle import train_oracle as to

    sd = dm.synth_state_dict(dm.unet_param_spec(CFG), salt=0)
    u, d = _model(sd)
    x_start, t, noise = _data(B, B)
    loss = float(d.p_losses(x_start, t, noise=noise))
    grads = d.model.grads()
    torch.set_num_threads(16)
    want_loss, want = to.loss_and_grads(sd, CFG, dm.make_schedule(1000, "linear"), x_start, t, noise)
    print(f"B={B}: loss", loss, want_loss)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    _compare(grads, want, f"B={B}")


def test_full_train_step_b64_vs_torch_adam():
    """One whole ``Trainer.train`` iteration at the benchmark shape (loss + backward, clip_grad_norm_(1.0), Adam, device
    re-pack) followed by a second loss on the UPDATED weights, against torch.optim.Adam on the oracle: the parameters after
    the step and the second loss agree, i.e. the device-side re-pack of every layout the B=64 plans use is right.  Also the
    asynchronous form (sync=False: 0-dim device tensors) returns the same numbers."""
    from oracle import train_oracle as to

    sd = dm.synth_state_dict(dm.unet_param_spec(CFG), salt=0)
    u, d = _model(sd)
    B, lr = 64, 2e-4
    x_start, t, noise = _data(B, 7)
    img = (x_start + 1) * 0.5
    loss0, norm0 = dm.train_step(d, [img], lr=lr, t=[t], noise=[noise], sync=False)
    assert isinstance(loss0, torch.Tensor) and loss0.is_cuda and loss0.dim() == 0
    loss0, norm0 = float(loss0), float(norm0)
    loss1 = float(d.p_losses(x_start, t, noise=noise))
    got = {k: v.cpu() for k, v in u.state_dict().items()}
    torch.set_num_threads(16)
    sched = dm.make_schedule(1000, "linear")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr, betas=(0.9, 0.99))
    xs = img * 2 - 1  # what train_step's normalize() forms
    l = to.p_losses(params, CFG, sched, xs, t, noise)
    l.backward()
    want_norm = float(torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0))
    opt.step()
    print("B=64 step: loss", loss0, float(l), "norm", norm0, want_norm)
    assert abs(loss0 - float(l)) <= 1e-5 * abs(float(l))
    assert abs(norm0 - want_norm) <= 2e-4 * want_norm
    # Adam's first step moves every element by ~lr * sign(g): compare the UPDATE, relative to its own size
    worst = ("", 0.0)
    for k in sd:
        dw, dg = params[k].detach() - sd[k], got[k] - sd[k]
        e = float((dw - dg).norm() / dw.norm().clamp_min(1e-30))
        worst = max(worst, (k, e), key=lambda v: v[1])
    print("worst parameter update", worst)
    assert worst[1] < 2e-3, worst  # sign flips of elements whose gradient is ~0 dominate; measured ~1e-4
    with torch.no_grad():
        want1 = float(to.p_losses({k: v.detach() for k, v in params.items()}, CFG, sched, xs, t, noise))
    print("loss after the step", loss1, want1)
    assert abs(loss1 - want1) <= 2e-5 * abs(want1)
