"""GPU parity of the training step against the REFERENCE's own autograd: loss and every
parameter gradient of ``DenoisingDiffusion.p_losses`` (DD/denoising_diffusion.py:823-889) for the fixtures of
tests/golden/make_golden_train.py, through ``dm_unet_loss_backward`` (C ABI).

Tolerances (fp32, rel-L2 of a gradient tensor against the reference's): 2e-4 per tensor checked through the digest
(norm, 8 random projections, first 256 elements, the whole tensor when small); loss 1e-5.  Measured values are ~1e-6."""
import pytest
import torch

import diffusion_models_amd as dm
from diffusion_models_amd.spec import UnetConfig

from conftest import check_grad_digest, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GRAD_TOL = 2e-4

CASES = {
    "small_d32": (UnetConfig(dim=32, dim_mults=(1, 2), channels=3), 41),
    "mid_d64": (UnetConfig(dim=64, dim_mults=(1, 2), channels=3), 42),
    "mid_d64_pred_x0": (UnetConfig(dim=64, dim_mults=(1, 2), channels=3), 42),
    "mid_d64_pred_v": (UnetConfig(dim=64, dim_mults=(1, 2), channels=3), 42),
    "full": (UnetConfig(), 0),
    "full_b8": (UnetConfig(), 0),
}


def _model(cfg, salt, objective, T):
    u = dm.Unet(dim=cfg.dim, dim_mults=cfg.dim_mults, channels=cfg.channels, device=DEV)
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=salt))
    side = 32 if len(cfg.dim_mults) == 4 else 16
    return dm.DenoisingDiffusion(u, image_size=side, timesteps=T, objective=objective).train()


@pytest.mark.parametrize("case", list(CASES))
def test_loss_and_all_gradients_vs_reference_autograd(golden_train, case):
    cfg, salt = CASES[case]
    b = golden_train[case]
    d = _model(cfg, salt, b["objective"], b["T"])
    x_start = b["img"] * 2 - 1
    assert rel_l2(d.q_sample(x_start, b["t"], b["noise"]).cpu(), b["x_noisy"]) < 1e-6
    loss = d.p_losses(x_start, b["t"], noise=b["noise"])
    print(case, "loss", float(loss), b["loss"])
    assert abs(float(loss) - b["loss"]) <= 1e-5 * abs(b["loss"])
    grads = d.model.grads()
    assert set(grads) == set(b["grads"])
    worst = ("", 0.0)
    for name, dg in b["grads"].items():
        g = grads[name].cpu()
        if "full" in dg:
            e = rel_l2(g, dg["full"]) if dg["norm"] > 0 else 0.0
            worst = max(worst, (name, e), key=lambda v: v[1])
        check_grad_digest(name, g, dg, GRAD_TOL)
    print(case, "worst fully-stored gradient", worst)


@pytest.mark.parametrize("case", ["offset", "immiscible"])
def test_noise_options_vs_reference_autograd(golden_train_noise, case):
    """Offset noise (:830-834) and immiscible diffusion (:805-817: the assignment on a device-side cdist + scipy on the
    host, q_sample on the re-assigned rows, the unpermuted noise as the target) against the reference's own loss and
    gradients."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    b = golden_train_noise[case]
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=41))
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=b["T"], immiscible=case == "immiscible").train()
    x_start = b["img"] * 2 - 1
    if case == "offset":
        noise = b["noise"].clone()
        loss = d.p_losses(x_start, b["t"], noise=noise, offset_noise_strength=b["strength"], offset_noise=b["offset"])
        assert torch.equal(noise, b["noise"])  # the caller's tensor is not modified
    else:
        assert d.noise_assignment(x_start, b["noise"]).tolist() == b["assign"].tolist()
        assert rel_l2(d.q_sample(x_start, b["t"], b["noise"]).cpu(), b["x_noisy"]) < 1e-6
        loss = d.p_losses(x_start, b["t"], noise=b["noise"])
    print(case, "loss", float(loss), b["loss"])
    assert abs(float(loss) - b["loss"]) <= 1e-5 * abs(b["loss"])
    grads = d.model.grads()
    for name, dg in b["grads"].items():
        check_grad_digest(name, grads[name].cpu(), dg, GRAD_TOL)
    if case == "offset":  # drawn on the device when not injected
        assert bool(torch.isfinite(d.p_losses(x_start, b["t"], offset_noise_strength=0.1)))


def test_forward_draws_t_and_noise(golden_train):
    """DenoisingDiffusion.forward (:892-899): img in [0, 1] -> normalize -> p_losses with random t / noise; with the
    reference's t and noise injected the loss is the reference's."""
    b = golden_train["forward_small_d32"]
    cfg, salt = CASES["small_d32"]
    d = _model(cfg, salt, "pred_noise", 1000)
    loss = d.p_losses(d.normalize(b["img"].to(DEV)), b["t"], noise=b["noise"])
    assert abs(float(loss) - b["loss"]) <= 1e-5 * abs(b["loss"])
    torch.manual_seed(3)
    l1 = float(d(b["img"]))
    torch.manual_seed(3)
    l2 = float(d(b["img"]))
    assert l1 == l2 and 0.0 < l1 < 10.0  # reproducible under torch.manual_seed, like the reference


def test_device_packers_match_host_packers():
    """The training loop re-packs every weight buffer on the device after an optimiser step (pack_kernels.hip); each device
    packer must reproduce its host packer bit for bit, for the forward layers and the input-gradient layers."""
    for cfg, salt in ((UnetConfig(), 0), (UnetConfig(dim=32, dim_mults=(1, 2), channels=3), 41)):
        d = _model(cfg, salt, "pred_noise", 1000)
        assert d.model.check_device_pack() == 0
        # after a step the model knows which buffers its shapes use: the grouped re-pack of exactly those (the rotated weights
        # of the input-gradient layers are read in place there) is compared with the per-buffer packers as well
        side = 32 if cfg.dim == 64 else 16
        float(d.p_losses(torch.rand(2, 3, side, side) * 2 - 1, torch.tensor([5, 700])))
        d.model.optimizer_step(lr=1e-4)
        assert d.model.check_device_pack() == 0


def _oracle_steps(cfg, sd, sched, batches, ts, noises, lr, n_steps, accumulate):
    """The reference loop on the CPU: autograd through the oracle, clip_grad_norm_(1.0), torch.optim.Adam."""
    from oracle import train_oracle as to

    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    opt = torch.optim.Adam(list(params.values()), lr=lr, betas=(0.9, 0.99))
    losses, norms = [], []
    for s in range(n_steps):
        opt.zero_grad()
        total = 0.0
        for i in range(accumulate):
            j = s * accumulate + i
            loss = to.p_losses(params, cfg, sched, batches[j] * 2 - 1, ts[j], noises[j]) / accumulate
            loss.backward()
            total += float(loss.detach())
        norms.append(float(torch.nn.utils.clip_grad_norm_(list(params.values()), 1.0)))
        opt.step()
        losses.append(total)
    return {k: v.detach() for k, v in params.items()}, losses, norms


@pytest.mark.parametrize("accumulate", [1, 2])
def test_training_steps_vs_torch_adam(accumulate):
    """Three iterations of Trainer.train's loop (:1164-1183) -- micro-batches, clip_grad_norm_(1.0), Adam(lr, (0.9, 0.99)) --
    on the device-resident state, against the same loop in torch on the CPU (oracle autograd): losses, gradient norms and
    every parameter after the last step.  Exercises the device-side re-packing between steps."""
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=42)
    n_steps, B, lr = 3, 4, 1e-3
    g = torch.Generator().manual_seed(77)
    n = n_steps * accumulate
    batches = [torch.rand((B, 3, 16, 16), generator=g) for _ in range(n)]
    ts = [torch.randint(0, 1000, (B,), generator=g) for _ in range(n)]
    noises = [torch.randn((B, 3, 16, 16), generator=g) for _ in range(n)]
    torch.set_num_threads(8)
    want, want_losses, want_norms = _oracle_steps(cfg, sd, dm.make_schedule(1000, "linear"), batches, ts, noises, lr, n_steps,
                                                  accumulate)
    d = _model(cfg, 42, "pred_noise", 1000)
    for s in range(n_steps):
        sl = slice(s * accumulate, (s + 1) * accumulate)
        loss, norm = dm.train_step(d, batches[sl], lr=lr, t=ts[sl], noise=noises[sl])
        print("step", s, "loss", loss, want_losses[s], "grad norm", norm, want_norms[s])
        assert abs(loss - want_losses[s]) <= 2e-4 * abs(want_losses[s])
        assert abs(norm - want_norms[s]) <= 1e-3 * want_norms[s]
    got = d.model.state_dict()
    worst = max((rel_l2(got[k].cpu(), want[k]), k) for k in want)
    print("worst parameter after", n_steps, "steps:", worst)
    # Adam divides by sqrt(v): where a gradient is ~0 the update direction is ill-conditioned, so the bound is on the
    # parameter (which moved by ~lr per step), not on the update
    assert worst[0] < 2e-4


def test_ema_and_sync_and_sampling_from_trained_weights():
    """EMA copy / lerp on the device, dm_unet_train_sync, and sampling with the trained weights: the handle that trained
    and a fresh handle loaded with its state_dict() give the same DDIM samples."""
    from oracle import sampler_oracle as so

    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    d = _model(cfg, 42, "pred_noise", 1000)
    ema = dm.EMA(d, beta=0.995, update_every=1, update_after_step=1)
    g = torch.Generator().manual_seed(5)
    hist = []
    for s in range(4):
        dm.train_step(d, [torch.rand((4, 3, 16, 16), generator=g)], lr=1e-3, ema=ema)
        hist.append({k: v.clone() for k, v in d.model.state_dict().items()})
    # ema_pytorch schedule: steps 0, 1 copy (step <= update_after_step), step 2 copies once more (first update after the
    # threshold initialises), step 3 lerps with decay(step=4) = 1 - (1 + 2)^(-2/3)
    decay = 1 - (1 + 2) ** (-2 / 3)
    got = d.model.state_dict(ema=True)
    k = "downs.0.0.block1.proj.weight"
    want = hist[2][k] * decay + hist[3][k] * (1 - decay)
    assert rel_l2(got[k], want) < 1e-6
    with pytest.raises(RuntimeError, match="dm_unet_train_sync"):
        d.ddim_sample((2, 3, 16, 16), sampling_timesteps=2, noise=so.NoiseStream(3))
    d.model.sync()
    a = d.ddim_sample((2, 3, 16, 16), sampling_timesteps=2, noise=so.NoiseStream(3))
    fresh = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    fresh.load_state_dict(d.model.state_dict())
    b = dm.DenoisingDiffusion(fresh, image_size=16, timesteps=1000).ddim_sample((2, 3, 16, 16), sampling_timesteps=2,
                                                                                noise=so.NoiseStream(3))
    assert torch.equal(a, b)
    e = ema.ema_model.ddim_sample((2, 3, 16, 16), sampling_timesteps=2, noise=so.NoiseStream(3))
    assert e.shape == a.shape and bool(torch.isfinite(e).all()) and not torch.equal(e, a)


def test_checkpoint_round_trip_and_torch_adam_reads_the_optimiser_state(tmp_path):
    """Trainer.save / Trainer.load (:1100-1133) on the device-resident state.
    (1) two iterations, save, a third: a fresh object that loads the file and runs the third iteration ends with
        bit-identical parameters, Adam moments and EMA copy (every kernel of the step is deterministic);
    (2) the file is the reference's layout: ``weights_only=True`` loads it, ``data['model']`` is
        ``DenoisingDiffusion.state_dict()``, ``data['ema']`` carries ``ema_model.*`` (what sampling.py:157-159 strips),
        and ``data['opt']`` loads into a real ``torch.optim.Adam`` over parameters in state-dict order, whose next step on
        the same gradients lands on the same parameters as dm_unet_optimizer_step;
    (3) loading a state dict into a handle whose parameters moved on the device does not compare against stale host copies."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    g = torch.Generator().manual_seed(11)
    batches = [torch.rand((4, 3, 16, 16), generator=g) for _ in range(3)]
    ts = [torch.randint(0, 1000, (4,), generator=g) for _ in range(3)]
    noises = [torch.randn((4, 3, 16, 16), generator=g) for _ in range(3)]
    lr = 1e-3

    def fresh():
        d = _model(cfg, 41, "pred_noise", 1000)
        return d, dm.EMA(d, beta=0.995, update_every=1, update_after_step=0)

    d, ema = fresh()
    for s in range(2):
        dm.train_step(d, [batches[s]], lr=lr, ema=ema, t=[ts[s]], noise=[noises[s]])
    path = tmp_path / "model-1.pt"
    dm.save_checkpoint(path, d, step=2, ema=ema, lr=lr)
    initial = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=41)
    trained = {k: v.clone() for k, v in d.model.state_dict().items()}
    # (2) the layout
    data = torch.load(str(path), map_location="cpu", weights_only=True)
    assert set(data) == {"step", "model", "opt", "ema", "scaler", "version"} and data["step"] == 2
    names = [n for n, _ in d.model.param_spec()]
    assert [k for k in data["model"] if k.startswith("model.")] == ["model." + n for n in names]
    assert "betas" in data["model"] and "loss_weight" in data["model"]
    assert dm.checkpoint.diffusion_state_dict_from_checkpoint(data).keys() == data["model"].keys()
    assert int(data["ema"]["step"]) == 2 and "ema_model.model." + names[0] in data["ema"]
    params = [torch.nn.Parameter(data["model"]["model." + n].clone()) for n in names]
    opt = torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99))
    opt.load_state_dict(data["opt"])
    assert all(float(opt.state[p]["step"]) == 2.0 for p in params)
    # third iteration on the handle that kept running
    dm.train_step(d, [batches[2]], lr=lr, ema=ema, t=[ts[2]], noise=[noises[2]])
    grads = d.model.grads()  # the (unclipped) gradients of the third iteration
    for p, n in zip(params, names):
        p.grad = grads[n].cpu().clone()
    torch.nn.utils.clip_grad_norm_(params, 1.0)
    opt.step()
    want = d.model.state_dict()
    worst = max((rel_l2(p.detach(), want[n].cpu()), n) for p, n in zip(params, names))
    print("torch.optim.Adam resumed from the file vs dm_unet_optimizer_step:", worst)
    assert worst[0] < 1e-6
    # (1) resume in a fresh object
    d2, ema2 = fresh()
    step, hyper = dm.load_checkpoint(path, d2, ema=ema2)
    assert step == 2 and abs(hyper["lr"] - lr) < 1e-12 and tuple(hyper["betas"]) == (0.9, 0.99)
    assert ema2.step == 2 and ema2.initted == ema.initted
    dm.train_step(d2, [batches[2]], lr=lr, ema=ema2, t=[ts[2]], noise=[noises[2]])
    for which, a, b in (("param", d.model.state_dict(), d2.model.state_dict()),
                        ("ema", d.model.state_dict(ema=True), d2.model.state_dict(ema=True)),
                        ("exp_avg", d.model._train_tensors(2), d2.model._train_tensors(2)),
                        ("exp_avg_sq", d.model._train_tensors(3), d2.model._train_tensors(3))):
        diff = [k for k in a if not torch.equal(a[k], b[k])]
        assert not diff, (which, diff[:3])
    assert d2.model._lib.dm_unet_adam_step(d2.model._handle, -1) == 3
    # the sampling scripts' use (sampling.py:157-159): an inference-only object, EMA(...).load_state_dict(data['ema']),
    # ema.ema_model.sample(...) -- the same images as the EMA model of the run that wrote the file
    from oracle import sampler_oracle as so

    u3 = dm.Unet(dim=cfg.dim, dim_mults=cfg.dim_mults, channels=3, device=DEV)
    u3.load_state_dict(initial)
    d3 = dm.DenoisingDiffusion(u3, image_size=16, timesteps=1000, sampling_timesteps=2)
    ema3 = dm.EMA(d3, beta=0.995, update_every=1)
    ema3.load_state_dict(data["ema"])
    ema3.ema_model.eval()
    ema_fresh = dm.EMA(d2, beta=0.995, update_every=1, update_after_step=0)
    ema_fresh.load_state_dict(data["ema"])  # training-mode object: into the device-resident EMA state
    a3 = ema3.ema_model.ddim_sample((2, 3, 16, 16), sampling_timesteps=2, noise=so.NoiseStream(4))
    b3 = ema_fresh.ema_model.ddim_sample((2, 3, 16, 16), sampling_timesteps=2, noise=so.NoiseStream(4))
    assert torch.equal(a3, b3) and bool(torch.isfinite(a3).all())
    # (3) the host copies of `d` still hold the initial weights; loading them back must reach the device all the same
    d.model.load_state_dict(initial)
    back = d.model.state_dict()
    assert all(torch.equal(back[k].cpu(), initial[k]) for k in initial)
    assert not torch.equal(trained[names[0]].cpu(), initial[names[0]])


def test_training_calls_on_different_streams_are_ordered():
    """loss + backward on one stream, the optimiser step (asynchronous: no host read-back requested by ema / re-pack) on
    another, three iterations: the handle orders the calls through its event, the parameters equal the single-stream run's
    bit for bit."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    g = torch.Generator().manual_seed(21)
    batches = [torch.rand((4, 3, 16, 16), generator=g) for _ in range(3)]
    ts = [torch.randint(0, 1000, (4,), generator=g) for _ in range(3)]
    noises = [torch.randn((4, 3, 16, 16), generator=g) for _ in range(3)]

    def run(streams):
        d = _model(cfg, 41, "pred_noise", 1000)
        for i in range(3):
            with torch.cuda.stream(streams[0]):
                d.p_losses(d.normalize(batches[i].to(DEV)), ts[i], noise=noises[i])
            with torch.cuda.stream(streams[1]):
                d.model.optimizer_step(lr=1e-3)
                d.model.ema_update(0.9, copy=i == 0)
        torch.cuda.synchronize()
        return d.model.state_dict(), d.model.state_dict(ema=True)

    s0 = torch.cuda.current_stream(DEV)
    want, want_ema = run((s0, s0))
    got, got_ema = run((torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)))
    assert all(torch.equal(want[k], got[k]) for k in want) and all(torch.equal(want_ema[k], got_ema[k]) for k in want_ema)


def test_training_reduces_the_loss_and_the_ema_model_samples():
    """End to end: 200 iterations of train_step (Adam 1e-3, clipping, EMA) on 16 fixed smooth images bring the denoising loss
    well under its starting level, the parameters stay finite, and the EMA model samples finite images in [0, 1]."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    d = _model(cfg, 41, "pred_noise", 1000)
    ema = dm.EMA(d, beta=0.99, update_every=5, update_after_step=20)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 16), torch.linspace(0, 1, 16), indexing="ij")
    g = torch.Generator().manual_seed(3)
    imgs = torch.stack([torch.stack([(yy * a + xx * (1 - a)), (yy * xx) ** b, (1 - yy) * a]) for a, b in
                        zip(torch.rand(16, generator=g).tolist(), (torch.rand(16, generator=g) + 0.5).tolist())]).float()
    torch.manual_seed(0)
    losses = [dm.train_step(d, [imgs], lr=1e-3, ema=ema)[0] for _ in range(200)]
    first, last = sum(losses[:10]) / 10, sum(losses[-10:]) / 10
    print("loss", first, "->", last)
    assert last < 0.35 * first and all(l == l for l in losses)
    assert all(bool(torch.isfinite(v).all()) for v in d.model.state_dict().values())
    out = ema.ema_model.ddim_sample((4, 3, 16, 16), sampling_timesteps=5)
    assert out.shape == (4, 3, 16, 16) and bool(torch.isfinite(out).all()) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0


def _block_shapes(cfg, side):
    """(C, H, W) of every Block output in the order Unet.forward runs them (two per ResnetBlock)."""
    dims = cfg.dims
    n = cfg.num_stages
    out = []
    h = side
    for i in range(n):
        out += [(dims[i], h, h)] * 4
        if i < n - 1:
            h //= 2
    out += [(dims[-1], h, h)] * 4  # mid_block1, mid_block2
    for j in range(n):
        out += [(dims[n - j], h, h)] * 4
        if j < n - 1:
            h *= 2
    out += [(dims[0], h, h)] * 2  # final_res_block
    return out


def test_dropout_training_step_with_the_same_masks_as_the_oracle():
    """Unet(dropout=0.1) as ddpm_cifar.yaml trains it: the library's Philox masks are exported (dm_op_dropout_mask) and
    injected into the oracle's Blocks; loss and every gradient must then agree.  Also: a new call draws new masks, the same
    seed reproduces a call, sampling (eval mode) ignores dropout."""
    import ctypes as C

    from diffusion_models_amd import _lib
    from oracle import train_oracle as to

    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=42)
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, dropout=0.1, device=DEV)
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000).train()
    p, seed, B = 0.1, 987654321, 4
    u.set_dropout_seed(seed)
    g = torch.Generator().manual_seed(9)
    x_start = torch.rand((B, 3, 16, 16), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 3, 16, 16), generator=g)
    loss0 = float(d.p_losses(x_start, t, noise=noise))
    grads = d.model.grads()
    lib = _lib.load()
    masks = []
    for k, (c, h, w) in enumerate(_block_shapes(cfg, 16)):
        m = torch.empty((B, h, w, c), device=DEV)
        _lib.check(lib.dm_op_dropout_mask(_lib.ptr(m), m.numel(), p, C.c_uint64(seed), C.c_uint64(0), k, None))
        masks.append(m.permute(0, 3, 1, 2).contiguous().cpu())
    keep = torch.cat([m.reshape(-1) for m in masks])
    frac = float((keep > 0).float().mean())
    assert abs(frac - 0.9) < 0.01 and bool(((keep == 0) | ((keep - 1 / 0.9).abs() < 1e-6)).all()), frac
    torch.set_num_threads(8)
    want_loss, want = to.loss_and_grads(sd, cfg, dm.make_schedule(1000, "linear"), x_start, t, noise, dropout_masks=masks)
    print("dropout loss", loss0, want_loss)
    assert abs(loss0 - want_loss) <= 1e-5 * abs(want_loss)
    worst = max((rel_l2(grads[k].cpu(), want[k]), k) for k in want)
    print("worst gradient with dropout", worst)
    assert worst[0] < GRAD_TOL
    loss1 = float(d.p_losses(x_start, t, noise=noise))      # next call: other masks
    assert loss1 != loss0
    u.set_dropout_seed(seed)
    assert float(d.p_losses(x_start, t, noise=noise)) == loss0  # same key: same masks
    from oracle import sampler_oracle as so

    plain = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    plain.load_state_dict(sd)
    a = d.ddim_sample((2, 3, 16, 16), sampling_timesteps=2, noise=so.NoiseStream(1))
    b = dm.DenoisingDiffusion(plain, image_size=16, timesteps=1000).ddim_sample((2, 3, 16, 16), sampling_timesteps=2,
                                                                                noise=so.NoiseStream(1))
    assert torch.equal(a, b)  # eval-mode sampling: dropout is the identity


def test_image_conditional_training_step_vs_oracle():
    """ImageConditionalDenoisingDiffusion.p_losses (denoising_diffusion_image_conditional.py:251-311): the condition image
    rides behind x through init_conv; loss and every gradient (incl. the 6-channel init_conv weight) against the oracle."""
    from oracle import train_oracle as to

    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, cond_channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=7)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, cond_channels=3, device=DEV)
    u.load_state_dict(sd)
    d = dm.ImageConditionalDenoisingDiffusion(u, image_size=16, timesteps=1000).train()
    g = torch.Generator().manual_seed(21)
    B = 4
    x_start = torch.rand((B, 3, 16, 16), generator=g) * 2 - 1
    cond = torch.rand((B, 3, 16, 16), generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 3, 16, 16), generator=g)
    loss = float(d.p_losses(x_start, t, noise=noise, cond=cond))
    want_loss, want = to.loss_and_grads(sd, cfg, dm.make_schedule(1000, "linear"), x_start, t, noise, cond=cond)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    got = d.model.grads()
    worst = max((rel_l2(got[k].cpu(), want[k]), k) for k in want)
    print("image-conditional worst gradient", worst)
    assert worst[0] < GRAD_TOL


@pytest.mark.parametrize("shape", [(24, 24), (20, 12), (8, 8), (16, 40)])
def test_training_step_at_other_image_sizes_vs_oracle(shape):
    """Loss and every gradient against the oracle's autograd on images that are not powers of two: 24x24 (stages 24, 12, 6),
    20x12 (non-square; a 5x3 bottleneck: odd maps take the direct weight-gradient kernel, even ones the Winograd-domain one)
    and 8x8 (a 2x2 bottleneck: several whole images per pixel block), 16x40 (rows wider than 32 pixels: the two-chunk form of
    init_conv's weight gradient), B = 3."""
    from oracle import train_oracle as to

    H, W = shape
    cfg = UnetConfig(dim=32, dim_mults=(1, 2, 4), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=11)
    u = dm.Unet(dim=32, dim_mults=(1, 2, 4), channels=3, device=DEV)
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=(H, W), timesteps=1000).train()
    g = torch.Generator().manual_seed(H * 100 + W)
    B = 3
    x_start = torch.rand((B, 3, H, W), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 3, H, W), generator=g)
    loss = float(d.p_losses(x_start, t, noise=noise))
    torch.set_num_threads(8)
    want_loss, want = to.loss_and_grads(sd, cfg, dm.make_schedule(1000, "linear"), x_start, t, noise)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    got = d.model.grads()
    worst = max((rel_l2(got[k].cpu(), want[k]), k) for k in want)
    print(shape, "worst gradient", worst)
    assert worst[0] < GRAD_TOL


@pytest.mark.parametrize("use_prediction", [False, True])
def test_self_conditioned_training_step_vs_oracle(use_prediction):
    """Unet(self_condition=True): p_losses conditions on zeros or (half of the iterations, :846-855) on the x_start a
    gradient-free forward pass predicts; loss and every gradient against the oracle for both branches."""
    from oracle import train_oracle as to

    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3, self_condition=True)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=32)
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, self_condition=True, device=DEV)
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000).train()
    g = torch.Generator().manual_seed(33)
    B = 4
    x_start = torch.rand((B, 3, 16, 16), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 3, 16, 16), generator=g)
    loss = float(d.p_losses(x_start, t, noise=noise, self_cond=use_prediction))
    torch.set_num_threads(8)
    want_loss, want = to.loss_and_grads(sd, cfg, dm.make_schedule(1000, "linear"), x_start, t, noise, self_cond=use_prediction)
    print("self-cond", use_prediction, "loss", loss, want_loss)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    got = d.model.grads()
    worst = max((rel_l2(got[k].cpu(), want[k]), k) for k in want)
    print("self-cond worst gradient", worst)
    assert worst[0] < GRAD_TOL


def test_latent_diffusion_training_loss():
    """LatentDiffusion.forward (latent_diffusion.py:51-56): encode the images with the frozen VQModel, then the base loss on
    the latents -- equal to calling the base class on the encoder's output with the same random draws."""
    from diffusion_models_amd.spec import DecoderConfig, EncoderConfig, encoder_param_spec

    ecfg = EncoderConfig(n_embed=64)
    vae = dm.VQModel(dict(ch=64, out_ch=3, in_channels=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=32,
                          z_channels=3, double_z=False), n_embed=64, embed_dim=3, device=DEV)
    vae.load_state_dict(dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(DecoderConfig()), salt=21))
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=42))
    ld = dm.LatentDiffusion(u, vae, latent_shape=(3, 16, 16), timesteps=1000).train()
    img = torch.rand((4, 3, 32, 32), generator=torch.Generator().manual_seed(1))
    torch.manual_seed(5)
    l1 = float(ld(img))
    torch.manual_seed(5)
    l2 = float(dm.DenoisingDiffusion.forward(ld, ld.encode(img)))
    assert l1 == l2 and 0.0 < l1 < 10.0
    g = ld.model.grad("init_conv.weight")
    assert bool(torch.isfinite(g).all()) and float(g.abs().sum()) > 0


@pytest.mark.parametrize("variant", ["concat", "cross_m1", "cross_m3", "cross_m3_px64"])
def test_text_conditional_training_step_vs_oracle(variant):
    """TextConditionalDenoisingDiffusion.p_losses (denoising_diffusion_text_conditional.py:476-542): the concat variant
    (text_proj -> cat(t, .) -> text_concat_proj) and the three CrossAttention layers around the bottleneck, with one pooled
    context token (what the reference's trainer passes: to_q / to_k receive exactly zero gradient, the softmax over a
    single key is constant) and with three tokens; loss and every gradient against the oracle.  ``px64``: a 32x32
    bottleneck -- 1024 query tokens through the tiled attention backward (mid_attn and the CrossAttention layers; the
    LDS-resident kernel holds about 300)."""
    from oracle import train_oracle as to

    cross = variant != "concat"
    px = 64 if variant.endswith("px64") else 16
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=cross)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=2 if cross else 3)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=cross, device=DEV)
    u.load_state_dict(sd)
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=px, timesteps=1000).train()
    g = torch.Generator().manual_seed(44)
    B = 4 if px == 16 else 2
    x_start = torch.rand((B, 3, px, px), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 3, px, px), generator=g)
    emb = torch.randn((B, 3, 512), generator=g) if "m3" in variant else torch.randn((B, 512), generator=g)
    loss = float(d.p_losses(x_start, t, emb, noise))
    torch.set_num_threads(8)
    want_loss, want = to.loss_and_grads(sd, cfg, dm.make_schedule(1000, "linear"), x_start, t, noise, text_emb=emb)
    print(variant, "loss", loss, want_loss)
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss)
    got = d.model.grads()
    scale = max(float(v.norm()) for v in want.values())
    for k in want:
        w = want[k]
        if float(w.norm()) < 1e-9 * scale:  # exact zeros in the reference (single context token): absolute check
            assert float(got[k].norm()) < 1e-6 * scale, k
        else:
            assert rel_l2(got[k].cpu(), w) < GRAD_TOL, (k, rel_l2(got[k].cpu(), w))


@pytest.mark.parametrize("cross", [False, True])
def test_caption_dropout_leaves_no_stale_text_gradients(cross):
    """p_losses with captions, then p_losses with text_emb=None on the same text-conditional U-Net (how caption dropout is
    trained; the reference's forward defaults text_emb to None): the second call visits no text parameter -- torch leaves
    their .grad None -- so their gradients must read zero, not the first call's values, and an optimiser step must leave
    those parameters where they were."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=cross)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=2 if cross else 3)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=cross, device=DEV)
    u.load_state_dict(sd)
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=16, timesteps=1000).train()
    g = torch.Generator().manual_seed(45)
    x_start = torch.rand((4, 3, 16, 16), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (4,), generator=g)
    noise = torch.randn((4, 3, 16, 16), generator=g)
    emb = torch.randn((4, 3, 512), generator=g) if cross else torch.randn((4, 512), generator=g)
    text_names = [k for k in sd if k.startswith(("text_", "cross_attn"))]
    assert text_names
    d.p_losses(x_start, t, emb, noise)
    assert any(float(u.grad(k).abs().sum()) > 0 for k in text_names)
    d.p_losses(x_start, t, None, noise)
    for k in text_names:
        assert float(u.grad(k).abs().sum()) == 0.0, k
    assert float(u.grad("init_conv.weight").abs().sum()) > 0
    before = {k: u.state_dict()[k].clone() for k in text_names}
    u.optimizer_step(lr=1e-3)
    after = u.state_dict()
    for k in text_names:
        assert torch.equal(before[k], after[k]), k
    assert not torch.equal(sd["init_conv.weight"], after["init_conv.weight"].cpu())


@pytest.mark.parametrize("objective", ["pred_noise", "pred_x0", "pred_v"])
def test_hybrid_loss_vs_reference_autograd(golden_hybrid, objective):
    """``DenoisingDiffusion(hybrid_loss=True)`` (:880-897) against the REFERENCE's loss.backward(): the loss and every
    gradient for a batch without t = 0; for a batch that holds t = 0 the reference's own answer is NaN (it divides by
    posterior_variance[0] = 0 before the mask multiplies) and so is ours."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    b = golden_hybrid["hybrid_" + objective]
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=41))
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=b["T"], objective=objective, hybrid_loss=True).train()
    x_start = b["img"] * 2 - 1
    loss = float(d.p_losses(x_start, b["t"], noise=b["noise"]))
    print(objective, "hybrid loss", loss, b["loss"], "without the KL term", b["loss_without_kl"])
    assert abs(loss - b["loss"]) <= 1e-5 * abs(b["loss"])
    grads = d.model.grads()
    for name, dg in b["grads"].items():
        check_grad_digest(name, grads[name].cpu(), dg, GRAD_TOL)
    z = golden_hybrid["hybrid_" + objective + "_t0"]
    loss0 = float(d.p_losses(x_start, z["t"], noise=b["noise"]))
    assert loss0 != loss0, loss0
    assert all(bool(torch.isnan(g).all()) for g in d.model.grads().values())
    # the asynchronous form returns the same number
    again = d.p_losses(x_start, b["t"], noise=b["noise"], sync=False)
    assert again.is_cuda and float(again) == loss


def test_hybrid_loss_text_conditional_and_with_dropout(golden_hybrid):
    """The text-conditional class's hybrid branch against the reference; and with dropout the KL term runs as a second,
    accumulating pass with its own masks (the reference's p_mean_variance call is a second forward pass): checked against the
    oracle with both passes' masks exported."""
    import ctypes as C

    from diffusion_models_amd import _lib
    from oracle import train_oracle as to

    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=True)
    b = golden_hybrid["hybrid_text_cross"]
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=True, device=DEV)
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=2))
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=16, timesteps=b["T"], hybrid_loss=True).train()
    loss = float(d.p_losses(b["img"] * 2 - 1, b["t"], b["emb"], b["noise"]))
    assert abs(loss - b["loss"]) <= 1e-5 * abs(b["loss"]), (loss, b["loss"])
    grads = d.model.grads()
    scale = max(dg["norm"] for dg in b["grads"].values())
    for name, dg in b["grads"].items():
        if dg["norm"] < 1e-9 * scale:
            assert float(grads[name].norm()) < 1e-6 * scale, name
        else:
            check_grad_digest(name, grads[name].cpu(), dg, GRAD_TOL)
    # dropout: two passes, two sets of masks
    cfg2 = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg2), salt=42)
    u2 = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, dropout=0.1, device=DEV)
    u2.load_state_dict(sd)
    d2 = dm.DenoisingDiffusion(u2, image_size=16, timesteps=1000, hybrid_loss=True).train()
    p, seed, B = 0.1, 424242, 4
    u2.set_dropout_seed(seed)
    g = torch.Generator().manual_seed(10)
    x_start = torch.rand((B, 3, 16, 16), generator=g) * 2 - 1
    t = torch.tensor([3, 400, 777, 999])
    noise = torch.randn((B, 3, 16, 16), generator=g)
    loss2 = float(d2.p_losses(x_start, t, noise=noise))
    got = d2.model.grads()
    lib = _lib.load()
    masks = [[], []]
    for call in (0, 1):
        for k, (c, h, w) in enumerate(_block_shapes(cfg2, 16)):
            m = torch.empty((B, h, w, c), device=DEV)
            _lib.check(lib.dm_op_dropout_mask(_lib.ptr(m), m.numel(), p, C.c_uint64(seed), C.c_uint64(call), k, None))
            masks[call].append(m.permute(0, 3, 1, 2).contiguous().cpu())
    torch.set_num_threads(8)
    want_loss, want = to.loss_and_grads(sd, cfg2, dm.make_schedule(1000, "linear"), x_start, t, noise, hybrid=True,
                                        dropout_masks=masks[0], kl_fwd_kw=dict(dropout_masks=masks[1]))
    print("hybrid + dropout: loss", loss2, want_loss)
    assert abs(loss2 - want_loss) <= 1e-5 * abs(want_loss)
    worst = max((rel_l2(got[k].cpu(), want[k]), k) for k in want)
    print("worst gradient", worst)
    assert worst[0] < GRAD_TOL


def test_non_ddpm_loss_weight_training_step_vs_oracle():
    """ddpm=False with the min-SNR clip (:535-549): the loss weight multiplies each sample's loss; against the oracle."""
    from oracle import train_oracle as to

    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=41)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, objective="pred_v", ddpm=False, min_snr_loss_weight=True).train()
    g = torch.Generator().manual_seed(12)
    x_start = torch.rand((4, 3, 16, 16), generator=g) * 2 - 1
    t = torch.tensor([0, 5, 500, 990])
    noise = torch.randn((4, 3, 16, 16), generator=g)
    loss = float(d.p_losses(x_start, t, noise=noise))
    sched = dm.make_schedule(1000, "linear", ddpm=False, objective="pred_v", min_snr_loss_weight=True)
    want_loss, want = to.loss_and_grads(sd, cfg, sched, x_start, t, noise, "pred_v")
    assert abs(loss - want_loss) <= 1e-5 * abs(want_loss), (loss, want_loss)
    got = d.model.grads()
    worst = max((rel_l2(got[k].cpu(), want[k]), k) for k in want)
    assert worst[0] < GRAD_TOL, worst


def test_bucketed_backward_gives_the_same_gradients():
    """Data-parallel gradient buckets (Unet.grad_buckets): the spans tile the flat gradient buffer in order, the full U-Net
    has several of them at the default 25 MB, and a backward pass that finishes its weight gradients bucket by bucket leaves
    the same gradients as the pass that runs them all at the end (other split tables: equal to rounding, not bit for bit)."""
    cfg = UnetConfig()
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=0)
    u = dm.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, device=DEV)
    u.load_state_dict(sd)
    d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000).train()
    g = torch.Generator().manual_seed(21)
    x_start = torch.rand((8, 3, 32, 32), generator=g) * 2 - 1
    t = torch.randint(0, 1000, (8,), generator=g)
    noise = torch.randn((8, 3, 32, 32), generator=g)
    loss_a = float(d.p_losses(x_start, t, noise=noise))
    ga = {k: v.clone() for k, v in d.model.grads().items()}
    spans = u.grad_buckets(enable=True)
    print("buckets (MB):", [round(4 * n / 2 ** 20, 1) for _, n in spans])
    assert len(spans) >= 3 and spans[0][0] == 0
    for (o0, n0), (o1, _) in zip(spans, spans[1:]):
        assert o0 + n0 == o1 and n0 > 0
    assert spans[-1][0] + spans[-1][1] == u.grads_flat().numel()
    loss_b = float(d.p_losses(x_start, t, noise=noise))
    gb = d.model.grads()
    assert loss_a == loss_b
    worst = max((rel_l2(gb[k], ga[k]), k) for k in ga)
    print("bucketed vs end-of-pass weight gradients, worst", worst)
    assert worst[0] < 1e-5
    # the spans are spans of grads_flat(), which holds every gradient exactly once (and zero padding)
    flat = u.grads_flat().double()
    total = sum(float(v.double().pow(2).sum()) for v in gb.values())
    assert abs(float(flat.pow(2).sum()) - total) <= 1e-9 * total
    assert all(float(flat[o:o + n].abs().sum()) > 0 for o, n in spans)
    u.grad_buckets(enable=False)
    assert float(d.p_losses(x_start, t, noise=noise)) == loss_a


def test_asynchronous_calls_keep_their_own_timesteps_and_coefficients():
    """sync=False returns before the GPU has run the call; the per-sample timesteps and schedule coefficients are host data
    handed over at call time.  Four calls with different (t, noise) enqueued back to back must give the losses the same calls
    give one at a time."""
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    d = _model(cfg, 42, "pred_noise", 1000)
    g = torch.Generator().manual_seed(31)
    x_start = torch.rand((8, 3, 16, 16), generator=g) * 2 - 1
    ts = [torch.randint(0, 1000, (8,), generator=g) for _ in range(4)]
    noises = [torch.randn((8, 3, 16, 16), generator=g).to(DEV) for _ in range(4)]
    xs = x_start.to(DEV)
    want = [float(d.p_losses(xs, t, noise=n)) for t, n in zip(ts, noises)]
    got = [d.p_losses(xs, t, noise=n, sync=False) for t, n in zip(ts, noises)]
    assert all(v.is_cuda for v in got)
    assert [float(v) for v in got] == want and len(set(want)) == 4
