"""GPU parity of the whole path (through dm_unet_forward / dm_sample / dm_decoder_forward) against
the golden vectors generated from the reference (tests/golden/*.pt) and against the oracle.

Tolerances (rel-L2, fp32): one U-Net forward <= 1e-4; whole sampling loops <= 1e-3
(BASELINE.json north_star); measured values are printed with -s."""
import pytest
import torch

import diffusion_models_amd as dm
from diffusion_models_amd.spec import DecoderConfig, EncoderConfig, UnetConfig, encoder_param_spec
from oracle import sampler_oracle as so

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FWD_TOL = 1e-4
LOOP_TOL = 1e-3


def build_unet(salt=0, **kw):
    u = dm.Unet(device=DEV, **kw)
    u.load_state_dict(dm.synth_state_dict(u.param_spec(), salt=salt))
    return u


@pytest.fixture(scope="module")
def small_unet():
    return build_unet(salt=1, dim=32, dim_mults=(1, 2), channels=3)


@pytest.fixture(scope="module")
def full_unet():
    return build_unet(salt=0, dim=64, dim_mults=(1, 2, 4, 8), channels=3)


def test_unet_small_forward(golden_blocks, small_unet):
    b = golden_blocks["unet_small"]
    y = small_unet(b["x"], b["t"]).cpu()
    err = rel_l2(y, b["y"])
    print("unet_small rel-L2", err)
    assert err < FWD_TOL


def test_unet_rejects_bad_size(small_unet):
    with pytest.raises(AssertionError):
        small_unet(torch.zeros(1, 3, 15, 16), torch.zeros(1, dtype=torch.long))


@pytest.mark.parametrize("key", ["learned", "random", "learned_dim8"])
def test_learned_sinusoidal_unet_forward(golden_r4, key):
    """Unet(learned_sinusoidal_cond=True / random_fourier_features=True) (denoising_diffusion.py:86-101, :271-278: the time
    embedding is cat(t, sin(t w 2 pi), cos(t w 2 pi)) with the parameter time_mlp.0.weights) against the reference's own
    forward.  Forward only: the reference's DenoisingDiffusion refuses such a U-Net (:456-457), and so do ours and train()."""
    b = golden_r4["unet_" + key]
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, **b["kw"])
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, device=DEV, **b["kw"])
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=51))
    err = rel_l2(u(b["x"].to(DEV), b["t"].to(DEV)).cpu(), b["y"])
    print("learned sinusoidal", key, err)
    assert err < FWD_TOL
    assert u.random_or_learned_sinusoidal_cond
    with pytest.raises(AssertionError):
        dm.DenoisingDiffusion(u, image_size=16)
    with pytest.raises(RuntimeError):
        u.train()


def test_per_stage_attention_heads(golden_r4):
    """Unet(attn_heads=(2, 4, 8)) (cast_tuple over the stages, denoising_diffusion.py:294; mid_attn: the last entry, :324):
    forward against the reference's output, then p_losses + backward against the reference's loss and gradient digests."""
    from conftest import check_grad_digest

    b = golden_r4["unet_stage_heads"]
    heads = tuple(b["heads"])
    cfg = UnetConfig(dim=32, dim_mults=(1, 2, 4), channels=3, attn_heads=heads)
    u = dm.Unet(dim=32, dim_mults=(1, 2, 4), channels=3, attn_heads=heads, device=DEV)
    u.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(cfg), salt=52))
    err = rel_l2(u(b["x"].to(DEV), b["t"].to(DEV)).cpu(), b["y"])
    print("per-stage heads forward", err)
    assert err < FWD_TOL
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=1000).train()
    loss = float(d.p_losses(b["img"] * 2 - 1, b["tt"], noise=b["noise"]))
    assert abs(loss - b["loss"]) <= 1e-4 * abs(b["loss"]), (loss, b["loss"])
    grads = d.model.grads()
    for name, dg in b["grads"].items():
        check_grad_digest(name, grads[name].cpu(), dg, 2e-4)
    with pytest.raises(NotImplementedError):
        dm.Unet(dim=32, dim_mults=(1, 2, 4), attn_dim_head=(32, 64, 32), device=DEV)


def test_unet_text_variants(golden_blocks):
    g = golden_blocks
    u = build_unet(salt=2, dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=True)
    for key in ("unet_text_cross", "unet_text_cross_m3"):
        b = g[key]
        err = rel_l2(u(b["x"], b["t"], text_emb=b["ctx"]).cpu(), b["y"])
        print(key, err)
        assert err < FWD_TOL
    u = build_unet(salt=3, dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=False)
    b = g["unet_text_concat"]
    err = rel_l2(u(b["x"], b["t"], text_emb=b["ctx"]).cpu(), b["y"])
    print("unet_text_concat", err)
    assert err < FWD_TOL


def test_unet_full_forward(golden_samplers, full_unet):
    for key in ("unet_full_32", "unet_full_64"):
        b = golden_samplers[key]
        err = rel_l2(full_unet(b["x"], b["t"]).cpu(), b["y"])
        print(key, err)
        assert err < FWD_TOL


def test_unet_latent_and_text_full(golden_samplers):
    u = build_unet(salt=0, dim=64, channels=4)
    b = golden_samplers["unet_latent4_32"]
    err = rel_l2(u(b["x"], b["t"]).cpu(), b["y"])
    print("latent4", err)
    assert err < FWD_TOL
    del u
    u = build_unet(salt=0, dim=64, text_condition=True, use_cross_attn=True)
    b = golden_samplers["unet_text_full_32"]
    err = rel_l2(u(b["x"], b["t"], text_emb=b["ctx"]).cpu(), b["y"])
    print("text_full", err)
    assert err < FWD_TOL


@pytest.mark.parametrize("use_graph", [False, True])
def test_small_samplers(golden_samplers, small_unet, use_graph):
    g = golden_samplers
    d = dm.DenoisingDiffusion(small_unet, image_size=16, timesteps=1000, use_graph=use_graph)
    b = g["small_ddim50"]
    y = d.ddim_sample(b["shape"], sampling_timesteps=b["S"], noise=so.NoiseStream(b["seed"])).cpu()
    print("small_ddim50", use_graph, rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL
    d.ddim_sampling_eta = 0.5
    b = g["small_ddim20_eta"]
    y = d.ddim_sample(b["shape"], sampling_timesteps=b["S"], noise=so.NoiseStream(b["seed"])).cpu()
    print("small_ddim20_eta", use_graph, rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL
    d50 = dm.DenoisingDiffusion(small_unet, image_size=16, timesteps=50, use_graph=use_graph)
    b = g["small_ddpm50_all"]
    y = d50.p_sample_loop(b["shape"], return_all_timesteps=True, noise=so.NoiseStream(b["seed"])).cpu()
    assert y.shape == b["y"].shape
    print("small_ddpm50_all", use_graph, rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL
    assert rel_l2(y[:, 1], b["y"][:, 1]) < 1e-4  # first iterate: one step of drift only


def test_small_ddpm1000(golden_samplers, small_unet):
    b = golden_samplers["small_ddpm1000"]
    d = dm.DenoisingDiffusion(small_unet, image_size=16, timesteps=1000)
    y = d.sample(batch_size=2, noise=so.NoiseStream(b["seed"])).cpu()
    print("small_ddpm1000", rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL


def test_full_samplers(golden_samplers, full_unet):
    g = golden_samplers
    d = dm.DenoisingDiffusion(full_unet, image_size=32, timesteps=1000, sampling_timesteps=50)
    assert d.is_ddim_sampling
    b = g["full_ddim50"]
    y = d.sample(batch_size=2, noise=so.NoiseStream(b["seed"])).cpu()
    print("full_ddim50", rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL
    d = dm.DenoisingDiffusion(full_unet, image_size=32, timesteps=1000)
    b = g["full_ddpm1000"]
    y = d.sample(batch_size=2, noise=so.NoiseStream(b["seed"])).cpu()
    print("full_ddpm1000", rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL


def test_sampling_properties_at_bench_size(full_unet):
    """Size-independent properties at the benchmark batch (B=256, 32x32, DDIM):
    determinism under a fixed Philox seed, range of the unnormalised output, and batch-shard
    independence (a sample does not depend on which other samples share its batch)."""
    d = dm.DenoisingDiffusion(full_unet, image_size=32, timesteps=1000, sampling_timesteps=50)
    a = d.ddim_sample((256, 3, 32, 32), sampling_timesteps=4, seed=7)
    b = d.ddim_sample((256, 3, 32, 32), sampling_timesteps=4, seed=7)
    assert torch.equal(a, b)
    assert torch.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0  # last DDIM step returns clamp(x0)
    c = d.ddim_sample((256, 3, 32, 32), sampling_timesteps=4, seed=8)
    assert not torch.equal(a, c)

    class Slice:  # injected noise: the same global rows whether sampled as 8 or as 2x4
        def __init__(self, lo, hi):
            self.s, self.lo, self.hi = so.NoiseStream(99), lo, hi

        def __call__(self, shape):
            return self.s((8,) + tuple(shape[1:]))[self.lo:self.hi]

    whole = d.ddim_sample((8, 3, 32, 32), sampling_timesteps=6, noise=Slice(0, 8))
    lo = d.ddim_sample((4, 3, 32, 32), sampling_timesteps=6, noise=Slice(0, 4))
    hi = d.ddim_sample((4, 3, 32, 32), sampling_timesteps=6, noise=Slice(4, 8))
    assert rel_l2(torch.cat((lo, hi)), whole) < 1e-5


def test_vae_decode(golden_vae):
    cfg = DecoderConfig()
    vae = dm.VQDecoder(dict(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=32,
                            z_channels=3), embed_dim=3, device=DEV)
    vae.load_state_dict(dm.synth_state_dict(dm.decoder_param_spec(cfg), salt=4))
    b = golden_vae["decode_cifar"]
    err = rel_l2(vae.decode(b["z"]).cpu(), b["y"])
    print("vae decode_cifar", err)
    assert err < FWD_TOL
    cfg2 = DecoderConfig(ch=32, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=(8,), resolution=16,
                         z_channels=4, embed_dim=4)
    vae2 = dm.VQDecoder(dict(ch=32, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=(8,),
                             resolution=16, z_channels=4), embed_dim=4, device=DEV)
    vae2.load_state_dict(dm.synth_state_dict(dm.decoder_param_spec(cfg2), salt=5))
    b = golden_vae["decode_attn3"]
    err = rel_l2(vae2.decode(b["z"]).cpu(), b["y"])
    print("vae decode_attn3", err)
    assert err < FWD_TOL


def test_vae_decode_is_bit_reproducible():
    """VQModel.decode several times on the same latents: bit-identical.  The GroupNorm statistics
    are a wave-shuffle + fixed-order two-stage reduction (csrc/vae_kernels.hip: group_sums_kernel), not atomics, so nothing
    depends on the order in which workgroups arrive; every other kernel of the decoder is deterministic by construction.
    Channel counts 64 (two groups per quad), 128 and 256 / 512 (butterfly over 2 / 4 lanes) are all on this path."""
    for ch, mult, res, z in ((64, (1, 2), 64, 4), (64, (1, 2, 4, 8), 32, 3)):
        cfg = DecoderConfig(ch=ch, ch_mult=mult, num_res_blocks=2, attn_resolutions=(), resolution=res, z_channels=z,
                            embed_dim=z)
        vae = dm.VQDecoder(dict(ch=ch, out_ch=3, ch_mult=mult, num_res_blocks=2, attn_resolutions=(), resolution=res,
                                z_channels=z), embed_dim=z, device=DEV)
        vae.load_state_dict(dm.synth_state_dict(dm.decoder_param_spec(cfg), salt=8))
        side = res // 2 ** (len(mult) - 1)
        lat = torch.randn((6, z, side, side), generator=torch.Generator().manual_seed(3)).to(DEV)
        a = vae.decode(lat)
        for _ in range(3):
            assert torch.equal(vae.decode(lat), a)
        assert bool(torch.isfinite(a).all())


def test_latent_diffusion_vs_oracle():
    """LatentDiffusion.sample = latent loop (no (x+1)/2) + decode (latent_diffusion.py:59-66), against the oracle's loop
    with unnormalize=False followed by the oracle's VQModel.decode (the reference's own LatentDiffusion.sample output
    at the config-4 shape is checked in tests/test_hip_configs.py)."""
    from oracle import unet_oracle as uo
    from oracle import vae_oracle as vo

    ucfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    usd = dm.synth_state_dict(dm.unet_param_spec(ucfg), salt=6)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, device=DEV)
    u.load_state_dict(usd)
    cfg = DecoderConfig()
    vsd = dm.synth_state_dict(dm.decoder_param_spec(cfg), salt=4)
    vae = dm.VQDecoder(dict(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=32,
                            z_channels=3), embed_dim=3, device=DEV)
    vae.load_state_dict(vsd)
    ld = dm.LatentDiffusion(u, vae, latent_shape=(3, 16, 16), timesteps=1000, sampling_timesteps=5)
    img = ld.sample(batch_size=2, noise=so.NoiseStream(3)).cpu()
    assert img.shape == (2, 3, 32, 32)
    with torch.inference_mode():
        lat = so.ddim_sample(lambda x, t: uo.unet_forward(usd, ucfg, x, t), dm.make_schedule(1000, "linear"),
                             (2, 3, 16, 16), so.NoiseStream(3), 5, unnormalize=False)
        want = vo.vq_decode(vsd, cfg, lat)
    assert lat.min() >= -1.0 and lat.max() <= 1.0  # identity unnormalize: clamp(x0) stays in [-1, 1]
    err = rel_l2(img, want)
    print("latent diffusion vs oracle", err)
    assert err < LOOP_TOL


@pytest.mark.parametrize("use_graph", [False, True])
def test_image_conditional(golden_imgcond, use_graph):
    """ImageConditionalDenoisingDiffusion (dm_sample_cond): condition concatenated in front of init_conv every step."""
    g = golden_imgcond
    u = build_unet(salt=7, dim=32, dim_mults=(1, 2), channels=3, cond_channels=3)
    b = g["unet_imgcond"]
    err = rel_l2(u(b["x"], b["t"], cond=b["cond"]).cpu(), b["y"])
    print("unet_imgcond", err)
    assert err < FWD_TOL
    d = dm.ImageConditionalDenoisingDiffusion(u, image_size=16, timesteps=50, use_graph=use_graph)
    b = g["imgcond_ddpm50"]
    cond, y = d.sample(batch_size=2, return_condition_image=True, cond=b["cond"], noise=so.NoiseStream(b["seed"]))
    assert torch.equal(cond.cpu(), b["cond"])
    print("imgcond_ddpm50", use_graph, rel_l2(y.cpu(), b["y"]))
    assert rel_l2(y.cpu(), b["y"]) < LOOP_TOL
    b = g["imgcond_ddim7"]
    y = d.ddim_sample(b["shape"], sampling_timesteps=b["S"], cond=b["cond"], noise=so.NoiseStream(b["seed"])).cpu()
    print("imgcond_ddim7", use_graph, rel_l2(y, b["y"]))
    assert rel_l2(y, b["y"]) < LOOP_TOL
    with pytest.raises(RuntimeError):
        d.sample(batch_size=2)  # no folder and no cond=


def test_edge_shapes_against_oracle():
    """Cases the golden files do not hold, checked against the (golden-pinned) oracle: a non-square image, a single
    image, a three-stage network wide enough for the Winograd / fused-attention kernels, cosine schedule, DDIM with
    all iterates returned."""
    from oracle import unet_oracle as uo

    cfg = UnetConfig(dim=64, dim_mults=(1, 2, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=9)
    u = dm.Unet(dim=64, dim_mults=(1, 2, 2), channels=3, device=DEV)
    u.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    for shape in ((1, 3, 32, 16), (3, 3, 8, 24)):
        x = torch.randn(shape, generator=g)
        t = torch.randint(0, 1000, (shape[0],), generator=g)
        with torch.inference_mode():
            want = uo.unet_forward(sd, cfg, x, t)
        err = rel_l2(u(x, t).cpu(), want)
        print("edge fwd", shape, err)
        assert err < FWD_TOL
    d = dm.DenoisingDiffusion(u, image_size=(16, 8), timesteps=200, beta_schedule="cosine", sampling_timesteps=9,
                              ddim_sampling_eta=0.3)
    got = d.ddim_sample((1, 3, 16, 8), return_all_timesteps=True, noise=so.NoiseStream(77)).cpu()
    with torch.inference_mode():
        want = so.ddim_sample(lambda x, t: uo.unet_forward(sd, cfg, x, t), dm.make_schedule(200, "cosine"), (1, 3, 16, 8),
                              so.NoiseStream(77), 9, eta=0.3, return_all_timesteps=True)
    assert got.shape == want.shape == (1, 10, 3, 16, 8)
    err = rel_l2(got, want)
    print("edge ddim cosine all-steps", err)
    assert err < LOOP_TOL


def test_text_conditional_loop():
    """TextConditionalDenoisingDiffusion.sample / p_sample_loop with explicit embeddings (the pickle lookup of random
    captions, denoising_diffusion_text_conditional.py:320-363, is host-side harness): text_emb rides through every step."""
    from oracle import unet_oracle as uo

    for cross in (True, False):
        cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=cross)
        sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=11)
        u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=cross, device=DEV)
        u.load_state_dict(sd)
        emb = torch.randn(2, 512, generator=torch.Generator().manual_seed(3))
        d = dm.TextConditionalDenoisingDiffusion(model=u, image_size=16, timesteps=40, sampling_timesteps=6)
        got = d.sample(batch_size=2, text_emb=emb, noise=so.NoiseStream(21)).cpu()
        model = lambda x, t: uo.unet_forward(sd, cfg, x, t, text_emb=emb)  # noqa: E731
        with torch.inference_mode():
            want = so.ddim_sample(model, dm.make_schedule(40, "linear"), (2, 3, 16, 16), so.NoiseStream(21), 6)
        err = rel_l2(got, want)
        print("text ddim", cross, err)
        assert err < LOOP_TOL
        got = d.p_sample_loop((2, 3, 16, 16), text_emb=emb, noise=so.NoiseStream(22)).cpu()
        with torch.inference_mode():
            want = so.p_sample_loop(model, dm.make_schedule(40, "linear"), (2, 3, 16, 16), so.NoiseStream(22))
        err = rel_l2(got, want)
        print("text ddpm", cross, err)
        assert err < LOOP_TOL


def test_vae_encode(golden_encoder):
    """dm_encoder_forward: Encoder + quant_conv against the reference's Encoder outputs pushed through the oracle's
    quant_conv, and the nearest-code quantiser against the oracle's restatement (same winner unless two codes tie to
    within fp32 rounding of the distances)."""
    from oracle import vae_oracle as vo

    cases = {
        "encode_cifar": (EncoderConfig(), 6),
        "encode_attn3": (EncoderConfig(ch=32, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=(8,), resolution=32,
                                       z_channels=4, embed_dim=4, n_embed=64), 7),
    }
    for name, (cfg, salt) in cases.items():
        sd = dm.synth_state_dict(encoder_param_spec(cfg), salt=salt)
        enc = dm.VQEncoder(dict(ch=cfg.ch, in_channels=3, ch_mult=cfg.ch_mult, num_res_blocks=cfg.num_res_blocks,
                                attn_resolutions=cfg.attn_resolutions, resolution=cfg.resolution,
                                z_channels=cfg.z_channels, double_z=False), cfg.embed_dim, cfg.n_embed, device=DEV)
        enc.load_state_dict(sd)
        b = golden_encoder[name]
        with torch.inference_mode():
            want_pre = torch.nn.functional.conv2d(b["h"], sd["quant_conv.weight"], sd["quant_conv.bias"])
            want_zq, want_idx = vo.vector_quantize(sd, want_pre)
        pre = enc.encode_to_prequant(b["x"]).cpu()
        err = rel_l2(pre, want_pre)
        print(name, "prequant", err)
        assert err < FWD_TOL
        zq, _, (_, _, idx) = enc.encode(b["x"])
        same = (idx.cpu() == want_idx).float().mean().item()
        print(name, "same code", same, "zq", rel_l2(zq.cpu(), want_zq))
        assert same > 0.99
        # every chosen code is a nearest one for the HIP pre-quant values
        flat = pre.permute(0, 2, 3, 1).reshape(-1, cfg.embed_dim).double()
        d = torch.cdist(flat, sd["quantize.embedding.weight"].double())
        assert torch.allclose(d.gather(1, idx.cpu()[:, None]).squeeze(1), d.min(dim=1).values, rtol=1e-4, atol=1e-5)


def test_image_conditional_latent_diffusion():
    """ImageConditionalLatentDiffusion (latent_diffusion_image_conditional.py): encode the condition image with the VQ
    model, run the image-conditional loop on latents, decode.  Checked against the oracle assembled from the same parts."""
    from oracle import unet_oracle as uo
    from oracle import vae_oracle as vo

    ecfg = EncoderConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, resolution=32, z_channels=3, embed_dim=3, n_embed=256)
    dcfg = DecoderConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, resolution=32, z_channels=3, embed_dim=3)
    vsd = dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=12)
    vae = dm.VQModel(dict(ch=32, out_ch=3, in_channels=3, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(),
                          resolution=32, z_channels=3, double_z=False), n_embed=256, embed_dim=3, device=DEV)
    vae.load_state_dict(vsd)
    ucfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, cond_channels=3)
    usd = dm.synth_state_dict(dm.unet_param_spec(ucfg), salt=13)
    u = dm.Unet(dim=32, dim_mults=(1, 2), channels=3, cond_channels=3, device=DEV)
    u.load_state_dict(usd)
    ld = dm.ImageConditionalLatentDiffusion(u, vae, latent_shape=(3, 16, 16), init_image_size=32, timesteps=30)
    cond = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(4))
    c, img = ld.sample(batch_size=2, return_condition_image=True, cond=cond, noise=so.NoiseStream(31))
    assert torch.equal(c.cpu(), cond) and img.shape == (2, 3, 32, 32)
    with torch.inference_mode():
        cl, _ = vo.vq_encode(vsd, ecfg, cond)
        lat = so.p_sample_loop(lambda x, t: uo.unet_forward(usd, ucfg, x, t, cond=cl), dm.make_schedule(30, "linear"),
                               (2, 3, 16, 16), so.NoiseStream(31), unnormalize=False)
        want = vo.vq_decode(vsd, dcfg, lat)
    err = rel_l2(img.cpu(), want)
    print("imgcond latent", err)
    assert err < LOOP_TOL


def test_state_dict_and_parameters_of_an_inference_handle():
    """``state_dict()`` / ``parameters()`` / ``named_parameters()`` without training mode: the values that were loaded, in the
    reference's parameter order; ``sum(p.numel() ...)`` as the training scripts print it."""
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=3)
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    with pytest.raises(RuntimeError):
        u.state_dict()
    u.load_state_dict(sd)
    got = u.state_dict()
    assert list(got) == [n for n, _ in dm.unet_param_spec(cfg)] and all(torch.equal(got[k].cpu(), sd[k]) for k in sd)
    n = sum(p.numel() for p in u.parameters())
    assert n == sum(v.numel() for v in sd.values()) and [k for k, _ in u.named_parameters()] == list(got)
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=10)
    assert sum(p.numel() for p in d.parameters()) == n
    # DenoisingDiffusion.state_dict(): 13 schedule buffers, then model.*; a second object loads it back
    dsd = d.state_dict()
    assert list(dsd)[:2] == ["betas", "alphas_cumprod"] and [k for k in dsd if k.startswith("model.")] == ["model." + k for k in got]
    u2 = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    d2 = dm.DenoisingDiffusion(u2, image_size=16, timesteps=10).load_state_dict(dsd)
    x = torch.randn((2, 3, 16, 16), generator=torch.Generator().manual_seed(1))
    t = torch.tensor([3, 7])
    assert torch.equal(d2.model(x, t), u(x, t))


def test_module_like_to_and_cuda():
    """``.to(device)`` / ``.cuda()`` / ``.eval()`` as the reference's scripts chain them: accepted on the object's own device,
    refused elsewhere (no CPU fallback, no silent move)."""
    u = dm.Unet(dim=64, dim_mults=(1, 2), channels=3, device=DEV)
    assert u.to(DEV) is u and u.to("cuda") is u and u.cuda() is u and u.to(torch.float32) is u
    d = dm.DenoisingDiffusion(u, image_size=16, timesteps=10)
    assert d.to(torch.device(DEV)).eval() is d and d.cuda(0) is d
    with pytest.raises(RuntimeError):
        u.to("cpu")
    with pytest.raises(RuntimeError):
        d.to("cuda:5")

