"""GPU parity of the five BASELINE.json configs AS WORKLOADS, and of the kernel plans the benchmark batch selects.

conv_plan / wino_plan choose split-K, images per workgroup, the wave grid and fused-vs-landing epilogues from the
workgroup count, i.e. from the batch: a B=2 golden does not exercise the plans a B=256 run takes.  Every case here goes
through the C ABI at the batch the config names and is compared with
  * the reference's own outputs (tests/golden/configs.pt, make_golden_configs.py) where the reference can produce them
    in the build container in seconds, and
  * the CPU oracle (pinned to those goldens by tests/test_oracle_golden.py) at the full batch.

Tolerances (rel-L2, fp32): one forward <= 1e-4; loops <= 1e-3 (BASELINE.json north_star)."""
import pytest
import torch

import diffusion_models_amd as dm
from diffusion_models_amd.spec import DecoderConfig, EncoderConfig, UnetConfig, encoder_param_spec
from oracle import sampler_oracle as so
from oracle import unet_oracle as uo
from oracle import vae_oracle as vo

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL = 1e-4
LOOP_TOL = 1e-3

FULL = UnetConfig()
CFG4_DEC = DecoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4,
                         embed_dim=4)
CFG4_ENC = EncoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4,
                         embed_dim=4, n_embed=256)
CFG4_DD = dict(ch=64, out_ch=3, in_channels=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64,
               z_channels=4, double_z=False)


def _unet(cfg: UnetConfig, salt=0):
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=salt)
    u = dm.Unet(dim=cfg.dim, dim_mults=cfg.dim_mults, channels=cfg.channels, text_condition=cfg.text_condition,
                use_cross_attn=cfg.use_cross_attn, cond_channels=cfg.cond_channels, device=DEV)
    u.load_state_dict(sd)
    return u, sd


@pytest.fixture(scope="module")
def full():
    return _unet(FULL)


def _rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


# ---- configs 1 / 2: 32x32 at the benchmark batch ----------------------------------------------------------------
def test_config2_b256_forward_and_ddim_vs_oracle(full):
    """B=256, 32x32: the Winograd / split-K / fused-norm plans of the measured run (bench.py), against the oracle."""
    u, sd = full
    torch.set_num_threads(max(1, min(64, torch.get_num_threads())))
    x = _rand((256, 3, 32, 32), 1)
    t = torch.randint(0, 1000, (256,), generator=torch.Generator().manual_seed(2))
    with torch.inference_mode():
        want = uo.unet_forward(sd, FULL, x, t)
    err = rel_l2(u(x, t).cpu(), want)
    print("B=256 forward vs oracle", err)
    assert err < FWD_TOL
    d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000, sampling_timesteps=50)
    got = d.ddim_sample((256, 3, 32, 32), sampling_timesteps=2, noise=so.NoiseStream(5)).cpu()
    with torch.inference_mode():
        want = so.ddim_sample(lambda xx, tt: uo.unet_forward(sd, FULL, xx, tt), dm.make_schedule(1000, "linear"),
                              (256, 3, 32, 32), so.NoiseStream(5), 2)
    err = rel_l2(got, want)
    print("B=256 DDIM-2 vs oracle", err)
    assert err < LOOP_TOL


def test_config1_b64_ddpm_steps_vs_oracle(full):
    """Config 1's shape (B=64, p_sample_loop) on the GPU: the first 3 of 1000 DDPM steps with all iterates."""
    u, sd = full
    d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000)
    got = d.p_sample_loop((64, 3, 32, 32), return_all_timesteps=True, noise=so.NoiseStream(6), max_steps=3).cpu()
    with torch.inference_mode():
        want = so.p_sample_loop(lambda xx, tt: uo.unet_forward(sd, FULL, xx, tt), dm.make_schedule(1000, "linear"),
                                (64, 3, 32, 32), so.NoiseStream(6), return_all_timesteps=True, num_steps=3)
    assert got.shape == want.shape == (64, 4, 3, 32, 32)
    err = rel_l2(got, want)
    print("B=64 DDPM first 3 steps vs oracle", err)
    assert err < LOOP_TOL


# ---- config 3: 64x64 DDPM, 32 and 8 images per GPU -------------------------------------------------------------
def test_config3_reference_goldens(golden_configs, full):
    u, _ = full
    b = golden_configs["unet_full_64_b2"]
    err = rel_l2(u(b["x"], b["t"]).cpu(), b["y"])
    print("unet_full_64_b2", err)
    assert err < FWD_TOL
    b = golden_configs["full64_ddpm50"]
    d = dm.DenoisingDiffusion(u, image_size=64, timesteps=b["T"])
    y = d.sample(batch_size=2, noise=so.NoiseStream(b["seed"])).cpu()
    err = rel_l2(y, b["y"])
    print("full64_ddpm50 (reference p_sample_loop)", err)
    assert err < LOOP_TOL


@pytest.mark.parametrize("B", [32, 8])
def test_config3_per_gpu_batches_vs_oracle(full, B):
    """64x64 at the per-GPU batches of the 8-way shard (global 256 -> 32, global 64 -> 8): forward + a short DDPM loop."""
    u, sd = full
    x = _rand((B, 3, 64, 64), 10 + B)
    t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(3))
    with torch.inference_mode():
        want = uo.unet_forward(sd, FULL, x, t)
    err = rel_l2(u(x, t).cpu(), want)
    print(f"64x64 B={B} forward vs oracle", err)
    assert err < FWD_TOL
    steps = 3 if B == 32 else 6
    d = dm.DenoisingDiffusion(u, image_size=64, timesteps=50)
    got = d.p_sample_loop((B, 3, 64, 64), noise=so.NoiseStream(7), max_steps=steps, return_all_timesteps=True).cpu()
    with torch.inference_mode():
        want = so.p_sample_loop(lambda xx, tt: uo.unet_forward(sd, FULL, xx, tt), dm.make_schedule(50, "linear"),
                                (B, 3, 64, 64), so.NoiseStream(7), num_steps=steps, return_all_timesteps=True)
    err = rel_l2(got, want)
    print(f"64x64 B={B} DDPM {steps} steps vs oracle", err)
    assert err < LOOP_TOL


# ---- config 4: latent 4x32x32 DDIM + VQModel.decode at 64x64 ---------------------------------------------------
@pytest.fixture(scope="module")
def cfg4():
    ucfg = UnetConfig(channels=4)
    u, usd = _unet(ucfg)
    vsd = dm.synth_state_dict(encoder_param_spec(CFG4_ENC) + dm.decoder_param_spec(CFG4_DEC), salt=14)
    vae = dm.VQModel(dict(CFG4_DD), n_embed=256, embed_dim=4, device=DEV)
    vae.load_state_dict(vsd)
    return u, usd, ucfg, vae, vsd


def test_config4_reference_goldens(golden_configs, cfg4):
    """LatentDiffusion.sample against the REFERENCE's own LatentDiffusion.sample (latent_diffusion.py:59-66)."""
    u, usd, ucfg, vae, vsd = cfg4
    b = golden_configs["decode_cfg4"]
    err = rel_l2(vae.decode(b["z"]).cpu(), b["y"])
    print("decode_cfg4", err)
    assert err < FWD_TOL
    ld = dm.LatentDiffusion(u, vae, latent_shape=(4, 32, 32), timesteps=1000, sampling_timesteps=6)
    b = golden_configs["latent4_ddim6"]
    lat = ld.ddim_sample(b["shape"], noise=so.NoiseStream(b["seed"])).cpu()
    err = rel_l2(lat, b["y"])
    print("latent4_ddim6", err)
    assert err < LOOP_TOL
    b = golden_configs["ldm_cfg4_ddim6"]
    img = ld.sample(batch_size=b["B"], noise=so.NoiseStream(b["seed"])).cpu()
    assert img.shape == (2, 3, 64, 64)
    err = rel_l2(img, b["y"])
    print("ldm_cfg4_ddim6 (reference LatentDiffusion.sample)", err)
    assert err < LOOP_TOL


def test_config4_b128_vs_oracle(cfg4):
    """B=128 (the config's batch): 2 DDIM steps of the latent loop and the decode to 64x64, against the oracle."""
    u, usd, ucfg, vae, vsd = cfg4
    ld = dm.LatentDiffusion(u, vae, latent_shape=(4, 32, 32), timesteps=1000, sampling_timesteps=200)
    lat = ld.ddim_sample((128, 4, 32, 32), sampling_timesteps=2, noise=so.NoiseStream(8))
    with torch.inference_mode():
        want = so.ddim_sample(lambda xx, tt: uo.unet_forward(usd, ucfg, xx, tt), dm.make_schedule(1000, "linear"),
                              (128, 4, 32, 32), so.NoiseStream(8), 2, unnormalize=False)
    err = rel_l2(lat.cpu(), want)
    print("latent B=128 DDIM-2 vs oracle", err)
    assert err < LOOP_TOL
    z = _rand((128, 4, 32, 32), 9)
    with torch.inference_mode():
        want = vo.vq_decode(vsd, CFG4_DEC, z)
    err = rel_l2(vae.decode(z).cpu(), want)
    print("decode B=128 -> 64x64 vs oracle", err)
    assert err < FWD_TOL


# ---- config 5: text / cross-attention U-Net at 64x64 ------------------------------------------------------------
@pytest.fixture(scope="module")
def text_full():
    return _unet(UnetConfig(text_condition=True, use_cross_attn=True))


def test_config5_reference_goldens(golden_configs, text_full):
    u, _ = text_full
    b = golden_configs["unet_text_full_64"]
    err = rel_l2(u(b["x"], b["t"], text_emb=b["ctx"]).cpu(), b["y"])
    print("unet_text_full_64", err)
    assert err < FWD_TOL
    b = golden_configs["text64_ddim4"]
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=64, timesteps=1000, sampling_timesteps=b["S"])
    y = d.sample(batch_size=2, text_emb=b["ctx"], noise=so.NoiseStream(b["seed"])).cpu()
    err = rel_l2(y, b["y"])
    print("text64_ddim4 (reference ddim_sample)", err)
    assert err < LOOP_TOL


def test_config5_b32_vs_oracle(text_full):
    u, sd = text_full
    cfg = UnetConfig(text_condition=True, use_cross_attn=True)
    emb = _rand((32, 512), 12)
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=64, timesteps=1000, sampling_timesteps=100)
    got = d.ddim_sample((32, 3, 64, 64), sampling_timesteps=2, text_emb=emb, noise=so.NoiseStream(13)).cpu()
    with torch.inference_mode():
        want = so.ddim_sample(lambda xx, tt: uo.unet_forward(sd, cfg, xx, tt, text_emb=emb),
                              dm.make_schedule(1000, "linear"), (32, 3, 64, 64), so.NoiseStream(13), 2)
    err = rel_l2(got, want)
    print("text 64x64 B=32 DDIM-2 vs oracle", err)
    assert err < LOOP_TOL


def test_text_conditional_latent_diffusion(cfg4):
    """TextConditionalLatentDiffusion (latent_diffusion_text_conditional.py:11-99; the reference class raises TypeError
    in its own __init__, see make_golden_configs.py): text loop on latents with identity (un)normalize, then decode --
    against the composition of the two halves the reference does pin."""
    _, _, _, vae, vsd = cfg4
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=4, text_condition=True, use_cross_attn=True)
    u, sd = _unet(cfg, salt=15)
    emb = _rand((2, 512), 14)
    tld = dm.TextConditionalLatentDiffusion(u, vae, latent_shape=(4, 32, 32), timesteps=1000, sampling_timesteps=5)
    img = tld.sample(batch_size=2, text_emb=emb, noise=so.NoiseStream(15)).cpu()
    assert img.shape == (2, 3, 64, 64)
    with torch.inference_mode():
        lat = so.ddim_sample(lambda xx, tt: uo.unet_forward(sd, cfg, xx, tt, text_emb=emb),
                             dm.make_schedule(1000, "linear"), (2, 4, 32, 32), so.NoiseStream(15), 5, unnormalize=False)
        want = vo.vq_decode(vsd, CFG4_DEC, lat)
    err = rel_l2(img, want)
    print("text LDM vs oracle composition", err)
    assert err < LOOP_TOL


# ---- multi-GPU invariants that one GPU can check ---------------------------------------------------------------
def test_sharded_seeded_sampling_equals_unsharded_bitwise(full):
    """Philox counters are GLOBAL element indices: shards [0,4) and [4,8) of one seed concatenate to the unsharded batch,
    bit for bit, in production mode (no injected noise) -- SURVEY.md 8(e)."""
    u, _ = full
    for d, kw in ((dm.DenoisingDiffusion(u, image_size=32, timesteps=1000, sampling_timesteps=5, ddim_sampling_eta=0.7), {}),
                  (dm.DenoisingDiffusion(u, image_size=32, timesteps=1000), dict(max_steps=5))):
        whole = d.sample(batch_size=8, seed=1234, **kw)
        assert torch.isfinite(whole).all()
        lo = d.sample(batch_size=4, seed=1234, sample_offset=0, **kw)
        hi = d.sample(batch_size=4, seed=1234, sample_offset=4, **kw)
        assert torch.equal(torch.cat((lo, hi)), whole)
        assert not torch.equal(lo, hi)
        r0 = d.sample(batch_size=3, seed=1234, sample_offset=0, **kw)
        r1 = d.sample(batch_size=3, seed=1234, sample_offset=3, **kw)
        r2 = d.sample(batch_size=2, seed=1234, sample_offset=6, **kw)  # ragged 3-way split
        assert torch.equal(torch.cat((r0, r1, r2)), whole)


def test_default_seed_follows_torch_manual_seed(full):
    u, _ = full
    d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000, sampling_timesteps=3)
    torch.manual_seed(77)
    a = d.sample(batch_size=2)
    b = d.sample(batch_size=2)
    torch.manual_seed(77)
    c = d.sample(batch_size=2)
    assert torch.equal(a, c) and not torch.equal(a, b)


def test_step_graph_is_captured_once_per_shape(full):
    """The instantiated hipGraph of a denoise step is cached on the handle: repeated sample() calls of one shape with
    different seeds replay it; only a new shape captures again."""
    u, _ = full
    d = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000, sampling_timesteps=4)
    d.sample(batch_size=4, seed=1)
    n0 = u.graph_captures
    outs = [d.sample(batch_size=4, seed=s) for s in (2, 3, 2)]
    assert u.graph_captures == n0
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])
    eager = dm.DenoisingDiffusion(u, image_size=32, timesteps=1000, sampling_timesteps=4, use_graph=False)
    assert torch.equal(eager.sample(batch_size=4, seed=2), outs[0])
    d.sample(batch_size=6, seed=1)
    assert u.graph_captures == n0 + 1


def test_in_place_weight_refresh():
    """Trainer.train samples from the EMA model after every update (denoising_diffusion.py:1190-1198): a second
    load_state_dict re-packs changed layers into the same device buffers (dm_unet_update_param / dm_unet_refresh);
    the result equals a freshly built handle bit for bit and the captured graph survives."""
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd_a = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=21)
    sd_b = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=22)
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(sd_a)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, sampling_timesteps=4)
    y_a = d.sample(batch_size=4, seed=5)
    caps = u.graph_captures
    u.load_state_dict(sd_b)  # full refresh
    y_b = d.sample(batch_size=4, seed=5)
    assert u.graph_captures == caps
    fresh = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    fresh.load_state_dict(sd_b)
    want_b = dm.DenoisingDiffusion(fresh, image_size=16, timesteps=1000, sampling_timesteps=4).sample(batch_size=4, seed=5)
    assert torch.equal(y_b, want_b) and not torch.equal(y_a, y_b)
    # partial refresh: one conv, one norm gain, one mlp, the attention of one stage
    sd_c = dict(sd_b)
    for k in ("downs.0.0.block1.proj.weight", "ups.1.1.block2.norm.g", "mid_block1.mlp.1.bias", "downs.1.2.to_qkv.weight",
              "final_conv.bias", "time_mlp.3.weight"):
        sd_c[k] = sd_a[k]
    u.load_state_dict(sd_c)
    fresh2 = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    fresh2.load_state_dict(sd_c)
    x = _rand((2, 3, 16, 16), 30)
    t = torch.tensor([3, 800])
    assert torch.equal(u(x, t), fresh2(x, t))
    assert not torch.equal(u(x, t), fresh(x, t))


# ---- checkpoints ---------------------------------------------------------------------------
def test_trainer_and_vae_checkpoints_load_into_the_hip_models(tmp_path):
    """A Trainer-style model-N.pt (EMA wrapper keys, denoising_diffusion.py:1100-1113 / sampling.py:157-159) and a
    Lightning-style VAE .ckpt load into DenoisingDiffusion / VQModel and sample like the oracle on the same weights."""
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    raw = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=31)
    ema = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=32)
    sched = dm.make_schedule(1000, "linear")
    wrap = lambda sd: {**{k: v for k, v in sched.items()}, **{"model." + k: v for k, v in sd.items()}}  # noqa: E731
    ckpt = {"step": 7, "model": wrap(raw), "opt": {}, "scaler": None, "version": "2.0.0",
            "ema": {**{"ema_model." + k: v for k, v in wrap(ema).items()},
                    **{"online_model." + k: v for k, v in wrap(raw).items()},
                    "initted": torch.tensor(True), "step": torch.tensor(7)}}
    path = tmp_path / "model-1.pt"
    torch.save(ckpt, path)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, device=DEV)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, sampling_timesteps=5)
    d.load_state_dict(dm.load_trainer_checkpoint(str(path)))
    got = d.sample(batch_size=2, noise=so.NoiseStream(41)).cpu()
    with torch.inference_mode():
        want = so.ddim_sample(lambda xx, tt: uo.unet_forward(ema, cfg, xx, tt), sched, (2, 3, 16, 16), so.NoiseStream(41), 5)
    err = rel_l2(got, want)
    print("trainer checkpoint (EMA weights) DDIM-5", err)
    assert err < LOOP_TOL
    d.load_state_dict(dm.load_trainer_checkpoint(str(path), prefer_ema=False))  # in-place refresh with the raw weights
    got = d.sample(batch_size=2, noise=so.NoiseStream(41)).cpu()
    with torch.inference_mode():
        want = so.ddim_sample(lambda xx, tt: uo.unet_forward(raw, cfg, xx, tt), sched, (2, 3, 16, 16), so.NoiseStream(41), 5)
    assert rel_l2(got, want) < LOOP_TOL

    ecfg = EncoderConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, resolution=32, z_channels=3, embed_dim=3, n_embed=64)
    dcfg = DecoderConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, resolution=32, z_channels=3, embed_dim=3)
    vsd = dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=33)
    lightning = {"epoch": 3, "global_step": 100, "pytorch-lightning_version": "2.5.1",
                 "state_dict": {**vsd, "loss.logvar": torch.zeros(()), "loss.discriminator.main.0.weight": torch.zeros(4, 3, 4, 4)}}
    vpath = tmp_path / "vae.ckpt"
    torch.save(lightning, vpath)
    vae = dm.VQModel(dict(ch=32, out_ch=3, in_channels=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(),
                          resolution=32, z_channels=3, double_z=False), n_embed=64, embed_dim=3, device=DEV)
    vae.load_state_dict(dm.load_vae_checkpoint(str(vpath)))
    z = _rand((2, 3, 16, 16), 42)
    with torch.inference_mode():
        want = vo.vq_decode(vsd, dcfg, z)
    assert rel_l2(vae.decode(z).cpu(), want) < FWD_TOL


# ---- p_sample of every wrapper ------------------------------------------------------------------------
def test_p_sample_of_every_wrapper_vs_oracle():
    sched = dm.make_schedule(1000, "linear")

    def check(name, got, want):
        for g, w, what in zip(got, want, ("pred_img", "x_start")):
            err = rel_l2(g.cpu(), w)
            print(name, what, err)
            assert err < FWD_TOL

    x = _rand((2, 3, 16, 16), 50)
    # plain
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    u, sd = _unet(cfg, salt=51)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000)
    for t in (500, 0):
        z = _rand(x.shape, 52)
        with torch.inference_mode():
            want = so.p_sample(lambda xx, tt: uo.unet_forward(sd, cfg, xx, tt), sched, x, t, z if t > 0 else None)
        check(f"plain t={t}", d.p_sample(x, t, noise=lambda s: z), want)
    # text: the reference's positional order is (x, t, text_emb)
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=True)
    u, sd = _unet(cfg, salt=53)
    emb = _rand((2, 512), 54)
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=16, timesteps=1000)
    for t in (500, 0):
        z = _rand(x.shape, 55)
        with torch.inference_mode():
            want = so.p_sample(lambda xx, tt: uo.unet_forward(sd, cfg, xx, tt, text_emb=emb), sched, x, t,
                               z if t > 0 else None)
        check(f"text t={t}", d.p_sample(x, t, emb, noise=lambda s: z), want)
    # image-conditional
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, cond_channels=3)
    u, sd = _unet(cfg, salt=56)
    cond = torch.rand((2, 3, 16, 16), generator=torch.Generator().manual_seed(57))
    d = dm.ImageConditionalDenoisingDiffusion(u, image_size=16, timesteps=1000)
    for t in (500, 0):
        z = _rand(x.shape, 58)
        with torch.inference_mode():
            want = so.p_sample(lambda xx, tt: uo.unet_forward(sd, cfg, xx, tt, cond=cond), sched, x, t,
                               z if t > 0 else None)
        check(f"imgcond t={t}", d.p_sample(x, t, cond, noise=lambda s: z), want)
