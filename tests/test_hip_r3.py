"""GPU parity against the round-3 fixtures generated from the reference (tests/golden/make_golden_r3.py):

* the shapes the reference's own LDM YAMLs produce -- 3x16x16 latents (ldm_cifar.yaml) and 3x8x8 latents
  (ldm_text_conditional_coco.yaml) through the 4-stage dim-64 U-Net, whose bottleneck is 2x2 / 1x1 (a 3x3 convolution
  on a 1x1 map, LinearAttention over 4 tokens, full Attention over 1 token + 4 memory keys), and the ch_mult (1,2,4,8)
  VQModel (decode 3x8x8 -> 3x64x64, encode_to_prequant 3x64x64 -> 3x8x8);
* objectives pred_x0 / pred_v and self-conditioning through both loops, eager and hipGraph replay.

Everything goes through the C ABI.  Tolerances (rel-L2, fp32): forward <= 1e-4, loops <= 1e-3."""
import pytest
import torch

import diffusion_models_amd as dm
from diffusion_models_amd.spec import DecoderConfig, EncoderConfig, UnetConfig, encoder_param_spec
from oracle import sampler_oracle as so
from oracle import unet_oracle as uo

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL = 1e-4
LOOP_TOL = 1e-3


def _unet(cfg: UnetConfig, salt):
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=salt)
    u = dm.Unet(dim=cfg.dim, dim_mults=cfg.dim_mults, channels=cfg.channels, self_condition=cfg.self_condition,
                text_condition=cfg.text_condition, use_cross_attn=cfg.use_cross_attn, device=DEV)
    u.load_state_dict(sd)
    return u, sd


@pytest.fixture(scope="module")
def full():
    return _unet(UnetConfig(), 0)


def test_ldm_cifar_shapes(golden_r3, full):
    """ldm_cifar.yaml: latents 3x16x16 and 3x8x8 through the 4-stage U-Net; LatentDiffusion.sample vs the reference's."""
    u, _ = full
    for side in (16, 8):
        b = golden_r3[f"unet_full_{side}"]
        err = rel_l2(u(b["x"], b["t"]).cpu(), b["y"])
        print(f"unet_full_{side}", err)
        assert err < FWD_TOL
    vae = dm.VQModel(dict(ch=64, out_ch=3, in_channels=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(),
                          resolution=32, z_channels=3, double_z=False), n_embed=8192, embed_dim=3, device=DEV)
    vae.load_state_dict(dm.synth_state_dict(encoder_param_spec(EncoderConfig(n_embed=8192)) +
                                            dm.decoder_param_spec(DecoderConfig()), salt=21))
    ld = dm.LatentDiffusion(u, vae, latent_shape=(3, 16, 16), timesteps=1000, sampling_timesteps=5)
    b = golden_r3["ldm_cifar_ddim5"]
    img = ld.sample(batch_size=b["B"], noise=so.NoiseStream(b["seed"])).cpu()
    assert img.shape == (2, 3, 32, 32) and ld.sample_shape() == (3, 32, 32)
    err = rel_l2(img, b["y"])
    print("ldm_cifar_ddim5 (reference LatentDiffusion.sample)", err)
    assert err < LOOP_TOL
    d8 = dm.DenoisingDiffusion(u, image_size=8, timesteps=1000, sampling_timesteps=5, auto_normalize=False)
    b = golden_r3["latent8_ddim5"]
    err = rel_l2(d8.ddim_sample(b["shape"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
    print("latent8_ddim5", err)
    assert err < LOOP_TOL
    d50 = dm.DenoisingDiffusion(u, image_size=8, timesteps=50, auto_normalize=False)
    b = golden_r3["latent8_ddpm50"]
    err = rel_l2(d50.p_sample_loop(b["shape"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
    print("latent8_ddpm50", err)
    assert err < LOOP_TOL


def test_ldm_cifar_shapes_at_training_batch(full):
    """The same latent shapes at the batch the YAMLs train / sample with (64 and 16): other tile / split-K plans."""
    u, sd = full
    cfg = UnetConfig()
    for B, side in ((64, 16), (16, 8)):
        x = torch.randn((B, 3, side, side), generator=torch.Generator().manual_seed(B))
        t = torch.randint(0, 1000, (B,), generator=torch.Generator().manual_seed(side))
        with torch.inference_mode():
            want = uo.unet_forward(sd, cfg, x, t)
        err = rel_l2(u(x, t).cpu(), want)
        print(f"B={B} {side}x{side} forward vs oracle", err)
        assert err < FWD_TOL


def test_ldm_coco_text_shapes(golden_r3):
    """ldm_text_conditional_coco.yaml: the text / cross-attention U-Net on 3x8x8 latents (bottleneck 1x1)."""
    u, _ = _unet(UnetConfig(text_condition=True, use_cross_attn=True), 0)
    b = golden_r3["unet_text_full_8"]
    err = rel_l2(u(b["x"], b["t"], text_emb=b["ctx"]).cpu(), b["y"])
    print("unet_text_full_8", err)
    assert err < FWD_TOL
    # the text-conditional Unet's positional order is (x, time, text_emb, x_self_cond)
    assert torch.equal(u(b["x"], b["t"], b["ctx"]), u(b["x"], b["t"], text_emb=b["ctx"]))
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=8, timesteps=1000, sampling_timesteps=4,
                                             auto_normalize=False)
    b = golden_r3["text8_ddim4"]
    err = rel_l2(d.ddim_sample(b["shape"], text_emb=b["ctx"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
    print("text8_ddim4", err)
    assert err < LOOP_TOL


def test_vq_coco(golden_r3):
    """VQModel with ch_mult (1,2,4,8), z_channels 3, n_embed 8192 at resolution 64 (the coco / edges2shoes VAEs)."""
    ecfg = EncoderConfig(ch=64, ch_mult=(1, 2, 4, 8), num_res_blocks=2, resolution=64, z_channels=3, embed_dim=3,
                         n_embed=8192)
    dcfg = DecoderConfig(ch=64, ch_mult=(1, 2, 4, 8), num_res_blocks=2, resolution=64, z_channels=3, embed_dim=3)
    vae = dm.VQModel(dict(ch=64, out_ch=3, in_channels=3, ch_mult=(1, 2, 4, 8), num_res_blocks=2, attn_resolutions=(),
                          resolution=64, z_channels=3, double_z=False), n_embed=8192, embed_dim=3, device=DEV)
    vae.load_state_dict(dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=22))
    b = golden_r3["vq_coco"]
    dec = vae.decode(b["z"]).cpu()
    assert dec.shape == (2, 3, 64, 64)
    err = rel_l2(dec, b["dec"])
    print("vq_coco decode", err)
    assert err < FWD_TOL
    err = rel_l2(vae.encode_to_prequant(b["x"]).cpu(), b["prequant"])
    print("vq_coco encode_to_prequant", err)
    assert err < FWD_TOL
    # VQModel.forward (autoencoder.py:123-128) = decode(encode(x)[0]), with the code indices on request
    dec, diff, ind = vae(b["x"], return_pred_indices=True)
    quant, _, (_, _, ind2) = vae.encode(b["x"])
    assert diff is None and torch.equal(ind, ind2) and torch.equal(dec, vae.decode(quant)) and len(vae(b["x"])) == 2


@pytest.mark.parametrize("use_graph", [False, True])
def test_objectives(golden_r3, use_graph):
    """pred_x0 / pred_v (DD/denoising_diffusion.py:614-624) through p_sample_loop and ddim_sample."""
    u, _ = _unet(UnetConfig(dim=64, dim_mults=(1, 2), channels=3), 31)
    for obj in ("pred_x0", "pred_v"):
        b = golden_r3[f"{obj}_ddpm50"]
        d = dm.DenoisingDiffusion(u, image_size=16, timesteps=b["T"], objective=obj, use_graph=use_graph)
        err = rel_l2(d.p_sample_loop(b["shape"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
        print(obj, "ddpm50", err)
        assert err < LOOP_TOL
        b = golden_r3[f"{obj}_ddim4"]
        d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, sampling_timesteps=b["S"], objective=obj,
                                  ddim_sampling_eta=b["eta"], use_graph=use_graph)
        err = rel_l2(d.ddim_sample(b["shape"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
        print(obj, "ddim4", err)
        assert err < LOOP_TOL


@pytest.mark.parametrize("use_graph", [False, True])
def test_self_conditioning(golden_r3, use_graph):
    """Unet(self_condition=True): the U-Net sees [x_start of the previous step | x] (:352-354, :657, :683)."""
    u, sd = _unet(UnetConfig(dim=64, dim_mults=(1, 2), channels=3, self_condition=True), 32)
    b = golden_r3["unet_selfcond"]
    assert rel_l2(u(b["x"], b["t"], b["x_self_cond"]).cpu(), b["y"]) < FWD_TOL
    assert rel_l2(u(b["x"], b["t"]).cpu(), b["y_none"]) < FWD_TOL
    b = golden_r3["selfcond_ddpm50"]
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=b["T"], use_graph=use_graph)
    err = rel_l2(d.p_sample_loop(b["shape"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
    print("selfcond_ddpm50", err)
    assert err < LOOP_TOL
    b = golden_r3["selfcond_ddim4"]
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, sampling_timesteps=b["S"], use_graph=use_graph)
    err = rel_l2(d.ddim_sample(b["shape"], noise=so.NoiseStream(b["seed"])).cpu(), b["y"])
    print("selfcond_ddim4", err)
    assert err < LOOP_TOL
    # p_sample with an explicit x_self_cond against the oracle's step
    x = torch.randn((2, 3, 16, 16), generator=torch.Generator().manual_seed(5))
    sc = torch.randn((2, 3, 16, 16), generator=torch.Generator().manual_seed(6)).clamp(-1, 1)
    z = torch.randn((2, 3, 16, 16), generator=torch.Generator().manual_seed(7))
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3, self_condition=True)
    with torch.inference_mode():
        want, want_x0 = so.p_sample(lambda xx, tt, s: uo.unet_forward(sd, cfg, xx, tt, s), dm.make_schedule(1000, "linear"),
                                    x, 400, z, x_self_cond=sc)
    got, got_x0 = d.p_sample(x, 400, sc, noise=lambda shape: z)
    assert rel_l2(got.cpu(), want) < FWD_TOL and rel_l2(got_x0.cpu(), want_x0) < FWD_TOL


def test_interpolate(golden_r3):
    """DenoisingDiffusion.interpolate (:786-803) against the reference's own run: q_sample of both images, the mix, the
    reverse loop from t - 1."""
    u, _ = _unet(UnetConfig(dim=64, dim_mults=(1, 2), channels=3), 31)
    b = golden_r3["interpolate"]
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=b["T"])
    got = d.interpolate(b["x1"], b["x2"], t=b["t"], lam=b["lam"], noise=so.NoiseStream(b["seed"])).cpu()
    err = rel_l2(got, b["y"])
    print("interpolate", err)
    assert err < LOOP_TOL


def test_prediction_helpers_and_guided_ddim(golden_guided):
    """model_predictions / p_mean_variance / q_posterior / predict_* as callable methods with per-sample timesteps (all three
    objectives) and ddim_sample_guided (DD/denoising_diffusion.py:570-636, :711-781) against the reference's own outputs."""
    u, _ = _unet(UnetConfig(dim=64, dim_mults=(1, 2), channels=3), 31)
    for obj in ("pred_noise", "pred_x0", "pred_v"):
        b = golden_guided[f"pred_{obj}"]
        d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, objective=obj)
        p = d.model_predictions(b["x"], b["t"])
        assert rel_l2(p.pred_noise.cpu(), b["pred_noise"]) < FWD_TOL and rel_l2(p.pred_x_start.cpu(), b["pred_x_start"]) < FWD_TOL
        p = d.model_predictions(b["x"], b["t"], clip_x_start=True, rederive_pred_noise=True)
        assert rel_l2(p.pred_noise.cpu(), b["pred_noise_clip"]) < FWD_TOL
        assert rel_l2(p.pred_x_start.cpu(), b["pred_x_start_clip"]) < FWD_TOL
        mean, var, logvar, xs = d.p_mean_variance(b["x"], b["t"])
        assert rel_l2(mean.cpu(), b["mean"]) < FWD_TOL and rel_l2(xs.cpu(), b["x_start"]) < FWD_TOL
        assert torch.equal(var.cpu(), b["var"]) and torch.equal(logvar.cpu(), b["logvar"]) and var.shape == (4, 1, 1, 1)
        # the pure elementwise helpers are bit-exact (same roundings as the reference's expressions)
        for name in ("predict_v", "predict_start_from_v", "predict_noise_from_start", "predict_start_from_noise"):
            got = getattr(d, name)(b["x"], b["t"], b["other"]).cpu()
            assert torch.equal(got, b[name]), (obj, name, rel_l2(got, b[name]))
        m2, v2, l2 = d.q_posterior(b["x_start"], b["x"], b["t"])
        assert rel_l2(m2.cpu(), b["mean"]) < 1e-6
    b = golden_guided["guided"]
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000, sampling_timesteps=b["S"], ddim_sampling_eta=b["eta"])
    y = d.ddim_sample_guided(b["shape"], guide=b["guide"], mask=b["mask"], noise=so.NoiseStream(b["seed"])).cpu()
    err = rel_l2(y, b["y"])
    print("ddim_sample_guided", err)
    assert err < LOOP_TOL
    y = d.ddim_sample_guided(b["shape"], noise=so.NoiseStream(b["seed_noguide"])).cpu()
    assert rel_l2(y, b["y_noguide"]) < LOOP_TOL
    assert bool(torch.isfinite(d.ddim_sample_guided(b["shape"], guide=b["guide"], mask=b["mask"])).all())  # device noise


def test_conditional_interpolate_and_vqmodel_ckpt_path(tmp_path):
    """``interpolate`` of the text- / image-conditional classes (positional order x1, x2, t, text_emb | cond, lam;
    denoising_diffusion_text_conditional.py:456-473, denoising_diffusion_image_conditional.py:232-249) against the oracle's
    loop with the condition closed over, and ``VQModel(ddconfig, lossconfig, n_embed, embed_dim, ckpt_path, ignore_keys)``
    (autoencoder.py:15-31, :78-91) loading a Lightning-style checkpoint."""
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3, text_condition=True)
    u, sd = _unet(cfg, 33)
    d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=16, timesteps=50)
    g = torch.Generator().manual_seed(9)
    x1, x2 = torch.rand((2, 3, 16, 16), generator=g) * 2 - 1, torch.rand((2, 3, 16, 16), generator=g) * 2 - 1
    emb = torch.randn((2, 512), generator=g)
    sched = dm.make_schedule(50, "linear")
    with torch.inference_mode():
        want = so.interpolate(lambda x, t: uo.unet_forward(sd, cfg, x, t, text_emb=emb), sched, x1, x2, 12, 0.3, so.NoiseStream(77))
    got = d.interpolate(x1, x2, 12, emb, 0.3, noise=so.NoiseStream(77)).cpu()
    err = rel_l2(got, want)
    print("text-conditional interpolate", err)
    assert err < LOOP_TOL
    icfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3, cond_channels=3)
    isd = dm.synth_state_dict(dm.unet_param_spec(icfg), salt=34)
    iu = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, cond_channels=3, device=DEV)
    iu.load_state_dict(isd)
    di = dm.ImageConditionalDenoisingDiffusion(iu, image_size=16, timesteps=50)
    cond = torch.rand((2, 3, 16, 16), generator=g)
    with torch.inference_mode():
        want = so.interpolate(lambda x, t: uo.unet_forward(isd, icfg, x, t, cond=cond), sched, x1, x2, 12, 0.3, so.NoiseStream(78))
    got = di.interpolate(x1, x2, 12, cond, 0.3, noise=so.NoiseStream(78)).cpu()
    err = rel_l2(got, want)
    print("image-conditional interpolate", err)
    assert err < LOOP_TOL
    # VQModel from a checkpoint file, reference argument order
    ddc = dict(ch=64, out_ch=3, in_channels=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=32,
               z_channels=3, double_z=False)
    vsd = dm.synth_state_dict(encoder_param_spec(EncoderConfig(n_embed=8192)) + dm.decoder_param_spec(DecoderConfig()), salt=21)
    extra = dict(vsd, **{"loss.discriminator.main.0.weight": torch.zeros(3)})
    path = tmp_path / "vq.ckpt"
    torch.save({"state_dict": extra}, str(path))
    a = dm.VQModel(ddc, None, 8192, 3, str(path), ["loss"], device=DEV)
    b = dm.VQModel(ddc, n_embed=8192, embed_dim=3, device=DEV)
    b.load_state_dict(vsd)
    z = torch.randn((2, 3, 16, 16), generator=g)
    assert torch.equal(a.decode(z), b.decode(z))


def test_vqmodel_with_groups_that_straddle_channel_quads():
    """``ch = 96`` gives GroupNorm(32) groups of 3, 6 and 12 channels: a 16-byte quad of a pixel row then belongs to two
    groups (a random-configuration sweep, tools/fuzz_vae.py, found the statistics kernel assuming otherwise).  Decode,
    encode_to_prequant and the code indices against the oracle."""
    from oracle import vae_oracle as vo

    common = dict(ch=96, ch_mult=(1, 2, 4), num_res_blocks=1, resolution=32, z_channels=4, embed_dim=4, attn_resolutions=(8,))
    ecfg, dcfg = EncoderConfig(n_embed=256, **common), DecoderConfig(**common)
    sd = dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=57)
    vae = dm.VQModel(dict(out_ch=3, in_channels=3, double_z=False, **{k: v for k, v in common.items() if k != "embed_dim"}),
                     n_embed=256, embed_dim=4, device=DEV)
    vae.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    z = torch.randn((2, 4, 8, 8), generator=g)
    img = torch.rand((2, 3, 32, 32), generator=g) * 2 - 1
    with torch.inference_mode():
        assert rel_l2(vae.decode(z).cpu(), vo.vq_decode(sd, dcfg, z)) < FWD_TOL
        assert rel_l2(vae.encode_to_prequant(img).cpu(), vo.vq_encode_to_prequant(sd, ecfg, img)) < FWD_TOL
        _, widx = vo.vq_encode(sd, ecfg, img)
    idx = vae.encode(img)[2][2].cpu().reshape(-1)
    assert float((idx == widx.reshape(-1)).float().mean()) > 0.99



def test_vqmodel_mid_attention_over_4096_tokens_with_32_channels():
    """A single-level VQModel (``ch_mult=(1,)``) at resolution 64 runs its mid-block attention over 64 x 64 = 4096 tokens of
    32 channels: the row kernel (C outside {64, 128, 256}) with more than 64 KB of LDS (tools/fuzz_vae.py found the old
    limit).  Decode and encode_to_prequant against the oracle."""
    from oracle import vae_oracle as vo

    common = dict(ch=32, ch_mult=(1,), num_res_blocks=1, resolution=64, z_channels=3, embed_dim=3, attn_resolutions=())
    ecfg, dcfg = EncoderConfig(n_embed=256, **common), DecoderConfig(**common)
    sd = dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=58)
    vae = dm.VQModel(dict(out_ch=3, in_channels=3, double_z=False, **{k: v for k, v in common.items() if k != "embed_dim"}),
                     n_embed=256, embed_dim=3, device=DEV)
    vae.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    z = torch.randn((1, 3, 64, 64), generator=g)
    img = torch.rand((1, 3, 64, 64), generator=g) * 2 - 1
    with torch.inference_mode():
        assert rel_l2(vae.decode(z).cpu(), vo.vq_decode(sd, dcfg, z)) < FWD_TOL
        assert rel_l2(vae.encode_to_prequant(img).cpu(), vo.vq_encode_to_prequant(sd, ecfg, img)) < FWD_TOL


def test_shape_mistakes_raise_before_any_launch():
    """Arguments whose shapes do not fit the model are refused on the host (the kernels index by the model's own channel
    count and the batch they are told): wrong channel count / noise shape / number of timesteps in ``p_losses``, a time
    vector of the wrong length in ``Unet.forward`` (a single entry broadcasts over the batch, as in the reference), sampler
    shapes the down-sampling factor does not divide."""
    u = dm.Unet(dim=16, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=3))
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 3, 8, 8), generator=g)
    t = torch.tensor([5, 900])
    with pytest.raises(RuntimeError, match="time has 3 entries"):
        u(x, torch.tensor([1, 2, 3]))
    with pytest.raises(RuntimeError, match="input channels"):
        u(torch.randn((2, 4, 8, 8), generator=g), t)
    one = u(x, torch.tensor([7]))
    assert torch.equal(one, u(x, torch.tensor([7, 7])))
    d = dm.DenoisingDiffusion(u, image_size=8, timesteps=1000)
    with pytest.raises(AssertionError, match="divisible"):
        d.p_sample_loop((1, 3, 8, 7))
    with pytest.raises(AssertionError, match="channels"):
        d.ddim_sample((1, 4, 8, 8), sampling_timesteps=2)
    d.train()
    with pytest.raises(RuntimeError, match="expected 3 channels"):
        d.p_losses(torch.randn((2, 4, 8, 8), generator=g), t)
    with pytest.raises(RuntimeError, match="do not match"):
        d.p_losses(x, t, noise=torch.randn((2, 3, 8, 4), generator=g))
    with pytest.raises(RuntimeError, match="do not match"):
        d.p_losses(x, torch.tensor([5]))
    assert float(d.p_losses(x, t)) > 0
