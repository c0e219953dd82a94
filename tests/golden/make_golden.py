#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (``/root/reference`` does not exist on the GPU
box):  ``python tests/golden/make_golden.py``.

What it does
------------
* imports the reference's own modules from ``/root/reference`` (read-only) with
  inert ``sys.modules`` stubs for the trainer-only imports that are not
  installed here (tensorboard, torchvision, ema_pytorch);
* instantiates the reference ``Unet`` / ``DenoisingDiffusion`` / text U-Net /
  VAE ``Decoder`` and loads *name-seeded synthetic weights*
  (``diffusion_models_amd.synth``) with ``strict=True`` -- which also proves our
  parameter spec equals the reference's ``state_dict()`` names and shapes;
* runs the reference on seeded inputs, with ``torch.randn`` / ``randn_like``
  inside the reference module redirected to one seeded CPU stream
  (``oracle.sampler_oracle.NoiseStream`` order), and stores inputs + outputs.

Only DATA is written (tensors, key lists); no reference source text.
"""
from __future__ import annotations

import importlib.machinery
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd.spec import DecoderConfig, UnetConfig  # noqa: E402
from oracle.sampler_oracle import NoiseStream  # noqa: E402


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, loader=None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

    _stub("torch.utils.tensorboard", SummaryWriter=_Dummy)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils")
    _stub("ema_pytorch", EMA=_Dummy)
    sys.path.insert(0, os.path.join(REF, "denoising-diffusion-pytorch"))
    sys.path.insert(0, os.path.join(REF, "latent-diffusion"))
    import denoising_diffusion.denoising_diffusion as dd
    import denoising_diffusion.denoising_diffusion_text_conditional as ddt
    import ldm.modules.diffusionmodules.model as ldm_model

    return dd, ddt, ldm_model


def save(name: str, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {name}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def seeded(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g)


class patched_noise:
    """Redirect randn / randn_like *as seen by one reference module* to a NoiseStream."""

    def __init__(self, module, seed):
        self.module, self.stream = module, NoiseStream(seed)

    def __enter__(self):
        real = self.module.torch
        stream = self.stream
        proxy = types.SimpleNamespace()

        class _T:  # attribute proxy over the torch module
            def __getattr__(_, k):
                if k == "randn":
                    return lambda shape, device=None, **kw: stream(shape)
                if k == "randn_like":
                    return lambda x, **kw: stream(x.shape)
                return getattr(real, k)

        self._real = real
        self.module.torch = _T()
        return self

    def __exit__(self, *exc):
        self.module.torch = self._real


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dd, ddt, ldm_model = import_reference()

    # ---- 1. schedules + ddim pairs ---------------------------------------------
    tiny = dd.Unet(dim=16, dim_mults=(1, 2), channels=3)
    sched = {}
    for name, kw in (("linear", {}), ("cosine", {}), ("sigmoid", {})):
        diff = dd.DenoisingDiffusion(tiny, image_size=16, timesteps=1000, beta_schedule=name)
        sched[name] = {k: v.clone() for k, v in diff.state_dict().items() if not k.startswith("model.")}
    diff250 = dd.DenoisingDiffusion(tiny, image_size=16, timesteps=250, beta_schedule="linear")
    sched["linear250"] = {k: v.clone() for k, v in diff250.state_dict().items() if not k.startswith("model.")}
    pairs = {}
    for T, S in ((1000, 50), (1000, 100), (1000, 200), (1000, 1000), (250, 7)):
        times = torch.linspace(-1, T - 1, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        pairs[f"{T}_{S}"] = list(zip(times[:-1], times[1:]))
    save("schedule.pt", {"sched": sched, "ddim_pairs": pairs})

    # ---- 2. state-dict key lists -------------------------------------------------
    keys = {}
    full_cfg = UnetConfig()
    ref_full = dd.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).eval()
    keys["unet_full"] = [(k, list(v.shape)) for k, v in ref_full.state_dict().items()]
    ref_text = ddt.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, text_condition=True, use_cross_attn=True).eval()
    keys["unet_text_cross"] = [(k, list(v.shape)) for k, v in ref_text.state_dict().items()]
    ref_text_cat = ddt.Unet(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=False).eval()
    keys["unet_text_concat_d32"] = [(k, list(v.shape)) for k, v in ref_text_cat.state_dict().items()]
    dcfg = DecoderConfig()
    ref_dec = ldm_model.Decoder(ch=64, out_ch=3, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=[],
                                in_channels=3, resolution=32, z_channels=3).eval()
    keys["decoder_cifar"] = [("decoder." + k, list(v.shape)) for k, v in ref_dec.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f)
    print("wrote state_dict_keys.json")

    # ---- 3. per-block goldens (small shapes, name-seeded weights) ---------------
    blocks = {}

    def load(mod, spec_prefix, sd_all):
        sd = {k[len(spec_prefix):]: v for k, v in sd_all.items() if k.startswith(spec_prefix)}
        mod.load_state_dict(sd, strict=True)
        return mod.eval()

    small_cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    small_sd = dm.synth_state_dict(dm.unet_param_spec(small_cfg), salt=1)
    ref_small = dd.Unet(dim=32, dim_mults=(1, 2), channels=3).eval()
    ref_small.load_state_dict(small_sd, strict=True)
    with torch.inference_mode():
        x = seeded((2, 32, 8, 8), 11)
        temb = seeded((2, 128), 12)
        blocks["rmsnorm"] = dict(x=x, g=small_sd["downs.0.0.block1.norm.g"],
                                 y=ref_small.downs[0][0].block1.norm(x))
        blocks["block_plain"] = dict(x=x, y=ref_small.downs[0][0].block2(x))
        ss = (seeded((2, 32, 1, 1), 13), seeded((2, 32, 1, 1), 14))
        blocks["block_ss"] = dict(x=x, scale=ss[0], shift=ss[1], y=ref_small.downs[0][0].block1(x, ss))
        blocks["resnet_same"] = dict(x=x, temb=temb, y=ref_small.downs[0][0](x, temb))
        xc = seeded((2, 96, 8, 8), 15)  # ups.0.0: 64+32 -> 64 with res_conv
        blocks["resnet_resconv"] = dict(x=xc, temb=temb, y=ref_small.ups[0][0](xc, temb))
        blocks["linattn"] = dict(x=x, y=ref_small.downs[0][2](x))
        x64 = seeded((2, 64, 4, 4), 16)
        blocks["fullattn"] = dict(x=x64, y=ref_small.mid_attn(x64))
        blocks["downsample"] = dict(x=x, y=ref_small.downs[0][3](x))
        blocks["upsample"] = dict(x=x64, y=ref_small.ups[0][3](x64))
        t = torch.tensor([0, 999], dtype=torch.long)
        blocks["time_mlp"] = dict(t=t, y=ref_small.time_mlp(t))
        blocks["sinusoid"] = dict(t=t, y=ref_small.time_mlp[0](t))
        xin = seeded((2, 3, 16, 16), 17)
        blocks["unet_small"] = dict(x=xin, t=t, y=ref_small(xin, t))
    # cross attention (text U-Net, small)
    tc_cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=True)
    tc_sd = dm.synth_state_dict(dm.unet_param_spec(tc_cfg), salt=2)
    ref_tc = ddt.Unet(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=True).eval()
    ref_tc.load_state_dict(tc_sd, strict=True)
    tcat_cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3, text_condition=True, use_cross_attn=False)
    tcat_sd = dm.synth_state_dict(dm.unet_param_spec(tcat_cfg), salt=3)
    ref_text_cat.load_state_dict(tcat_sd, strict=True)
    with torch.inference_mode():
        xf = seeded((2, 16, 64), 21)
        ctx1 = seeded((2, 512), 22)
        ctx3 = seeded((2, 3, 512), 23)
        blocks["cross_m1"] = dict(x=xf, ctx=ctx1, y=ref_tc.cross_attn(xf, ctx1))
        blocks["cross_m3"] = dict(x=xf, ctx=ctx3, y=ref_tc.cross_attn(xf, ctx3))
        blocks["unet_text_cross"] = dict(x=xin, t=t, ctx=ctx1, y=ref_tc(xin, t, text_emb=ctx1))
        blocks["unet_text_cross_m3"] = dict(x=xin, t=t, ctx=ctx3, y=ref_tc(xin, t, text_emb=ctx3))
        blocks["unet_text_concat"] = dict(x=xin, t=t, ctx=ctx1, y=ref_text_cat(xin, t, text_emb=ctx1))
    save("blocks.pt", blocks)

    # ---- 4. VAE decoder ------------------------------------------------------------
    vae = {}
    dsd = dm.synth_state_dict(dm.decoder_param_spec(dcfg), salt=4)
    ref_dec.load_state_dict({k[len("decoder."):]: v for k, v in dsd.items() if k.startswith("decoder.")}, strict=True)
    with torch.inference_mode():
        z = seeded((2, 3, 16, 16), 31)
        import torch.nn.functional as F

        q = F.conv2d(z, dsd["post_quant_conv.weight"], dsd["post_quant_conv.bias"])
        vae["decode_cifar"] = dict(z=z, y=ref_dec(q))
        xb = seeded((2, 128, 8, 8), 32)
        vae["resblock"] = dict(x=xb, y=ref_dec.mid.block_1(xb, None))
        vae["attnblock"] = dict(x=xb, y=ref_dec.mid.attn_1(xb))
        xb2 = seeded((2, 128, 8, 8), 33)
        vae["resblock_nin"] = dict(x=xb2, y=ref_dec.up[0].block[0](xb2, None))
    # a decoder with attention inside an up level and 3 levels
    dcfg2 = DecoderConfig(ch=32, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=(8,), resolution=16,
                          z_channels=4, embed_dim=4)
    ref_dec2 = ldm_model.Decoder(ch=32, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=1, attn_resolutions=[8],
                                 in_channels=3, resolution=16, z_channels=4).eval()
    dsd2 = dm.synth_state_dict(dm.decoder_param_spec(dcfg2), salt=5)
    ref_dec2.load_state_dict({k[len("decoder."):]: v for k, v in dsd2.items() if k.startswith("decoder.")}, strict=True)
    with torch.inference_mode():
        z2 = seeded((2, 4, 4, 4), 34)
        q2 = F.conv2d(z2, dsd2["post_quant_conv.weight"], dsd2["post_quant_conv.bias"])
        vae["decode_attn3"] = dict(z=z2, y=ref_dec2(q2))
    save("vae.pt", vae)

    # ---- 5. samplers: small config, full loops with injected noise --------------------
    samplers = {}
    diff_small = dd.DenoisingDiffusion(ref_small, image_size=16, timesteps=1000).eval()
    with patched_noise(dd, 101):
        samplers["small_ddim50"] = dict(seed=101, shape=(2, 3, 16, 16), S=50, eta=0.0,
                                        y=diff_small.ddim_sample((2, 3, 16, 16), sampling_timesteps=50))
    diff_small.ddim_sampling_eta = 0.5
    with patched_noise(dd, 102):
        samplers["small_ddim20_eta"] = dict(seed=102, shape=(2, 3, 16, 16), S=20, eta=0.5,
                                            y=diff_small.ddim_sample((2, 3, 16, 16), sampling_timesteps=20))
    diff_small.ddim_sampling_eta = 0.0
    with patched_noise(dd, 103):
        samplers["small_ddpm1000"] = dict(seed=103, shape=(2, 3, 16, 16),
                                          y=diff_small.p_sample_loop((2, 3, 16, 16)))
    diff_small_t50 = dd.DenoisingDiffusion(ref_small, image_size=16, timesteps=50).eval()
    with patched_noise(dd, 104):
        y_all = diff_small_t50.p_sample_loop((1, 3, 16, 16), return_all_timesteps=True)
        samplers["small_ddpm50_all"] = dict(seed=104, shape=(1, 3, 16, 16), T=50, y=y_all)

    # ---- 6. full config (35.7 M params, weights regenerated by name) ------------------
    full_sd = dm.synth_state_dict(dm.unet_param_spec(full_cfg), salt=0)
    ref_full.load_state_dict(full_sd, strict=True)
    full = {}
    with torch.inference_mode():
        xin = seeded((2, 3, 32, 32), 41)
        tt = torch.tensor([17, 903], dtype=torch.long)
        full["unet_full_32"] = dict(x=xin, t=tt, y=ref_full(xin, tt))
        xin64 = seeded((1, 3, 64, 64), 42)
        full["unet_full_64"] = dict(x=xin64, t=tt[:1], y=ref_full(xin64, tt[:1]))
    diff_full = dd.DenoisingDiffusion(ref_full, image_size=32, timesteps=1000).eval()
    with patched_noise(dd, 201):
        full["full_ddim50"] = dict(seed=201, shape=(2, 3, 32, 32), S=50, eta=0.0,
                                   y=diff_full.ddim_sample((2, 3, 32, 32), sampling_timesteps=50))
    with patched_noise(dd, 202):
        full["full_ddpm1000"] = dict(seed=202, shape=(2, 3, 32, 32), y=diff_full.p_sample_loop((2, 3, 32, 32)))
    # latent config L: channels=4 on 32x32
    lat_cfg = UnetConfig(channels=4)
    ref_lat = dd.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=4).eval()
    ref_lat.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(lat_cfg), salt=0), strict=True)
    with torch.inference_mode():
        xl = seeded((2, 4, 32, 32), 43)
        full["unet_latent4_32"] = dict(x=xl, t=tt, y=ref_lat(xl, tt))
    # text config T at 64x64 is 4x the work; pin at 32x32 with the full widths
    ref_text.load_state_dict(dm.synth_state_dict(dm.unet_param_spec(
        UnetConfig(text_condition=True, use_cross_attn=True)), salt=0), strict=True)
    with torch.inference_mode():
        ctx = seeded((2, 512), 44)
        full["unet_text_full_32"] = dict(x=xin, t=tt, ctx=ctx, y=ref_text(xin, tt, text_emb=ctx))
    samplers.update(full)
    save("samplers.pt", samplers)


if __name__ == "__main__":
    main()
