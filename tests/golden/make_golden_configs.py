#!/usr/bin/env python3
"""Golden vectors for BASELINE configs 3, 4 and 5 as WORKLOADS, generated from the REFERENCE itself.

Run in the build container only:  ``python tests/golden/make_golden_configs.py``  ->  ``tests/golden/configs.pt``.

* config 3: the 35.7 M-parameter U-Net at 64x64 through the reference's own ``p_sample_loop`` (a 50-step schedule,
  B=2, injected noise);
* config 4: ``Unet(channels=4)`` on 4x32x32 latents through the reference's ``LatentDiffusion.sample`` (DDIM) with the
  reference ``VQModel`` built from ``Decoder(ch=64, ch_mult=(1,2), z_channels=4, resolution=64)``, and
  that decoder alone;
* config 5: the full-width text / cross-attention U-Net at 64x64 -- one forward and the reference's
  ``TextConditionalDenoisingDiffusion.ddim_sample`` with the caption embeddings fixed (the pickle lookup of random
  captions is replaced on the instance by the stored embeddings).

Imports follow make_golden.py (inert stubs for trainer-only modules); the LDM wrappers additionally need
``pytorch_lightning`` (``LightningModule`` := ``nn.Module``) and taming's ``VectorQuantizer2`` (never executed by
``decode``) -- SURVEY.md 8(c).  Only DATA is written.
"""
from __future__ import annotations

import os
import sys
import tempfile

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_golden import REF, _stub, import_reference, patched_noise, save, seeded  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd.spec import DecoderConfig, EncoderConfig, UnetConfig, encoder_param_spec  # noqa: E402


def import_ldm():
    class _Quant(nn.Module):  # taming's VectorQuantizer2: constructed by VQModel.__init__, not run by decode()
        def __init__(self, n_e, e_dim, beta=0.25, remap=None, sane_index_shape=False):
            super().__init__()
            self.embedding = nn.Embedding(n_e, e_dim)

    pl = _stub("pytorch_lightning", LightningModule=nn.Module)
    pl.__version__ = "2.5.1"
    _stub("taming")
    _stub("taming.modules")
    _stub("taming.modules.vqvae")
    _stub("taming.modules.vqvae.quantize", VectorQuantizer2=_Quant)
    import ldm.models.autoencoder as ae
    import ldm.models.latent_diffusion as ld
    return ae, ld


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dd, ddt, ldm_model = import_reference()
    ae, ld = import_ldm()
    out = {}
    tt = torch.tensor([17, 903], dtype=torch.long)

    # ---- config 3: 64x64 DDPM loop on the full U-Net ------------------------------------------------
    full_sd = dm.synth_state_dict(dm.unet_param_spec(UnetConfig()), salt=0)
    ref_full = dd.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3).eval()
    ref_full.load_state_dict(full_sd, strict=True)
    d50 = dd.DenoisingDiffusion(ref_full, image_size=64, timesteps=50).eval()
    with patched_noise(dd, 301):
        out["full64_ddpm50"] = dict(seed=301, shape=(2, 3, 64, 64), T=50, y=d50.p_sample_loop((2, 3, 64, 64)))
    with torch.inference_mode():
        x2 = seeded((2, 3, 64, 64), 45)
        out["unet_full_64_b2"] = dict(x=x2, t=tt, y=ref_full(x2, tt))
    del ref_full

    # ---- config 4: latent U-Net (channels=4) + VQModel(decoder res 64, z 4) ---------------------------
    ddconfig = dict(double_z=False, z_channels=4, resolution=64, in_channels=3, out_ch=3, ch=64, ch_mult=[1, 2],
                    num_res_blocks=2, attn_resolutions=[], dropout=0.0)
    vq = ae.VQModel(ddconfig=ddconfig, lossconfig={"target": "torch.nn.Identity"}, n_embed=256, embed_dim=4).eval()
    dcfg = DecoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4,
                         embed_dim=4)
    ecfg = EncoderConfig(ch=64, ch_mult=(1, 2), num_res_blocks=2, attn_resolutions=(), resolution=64, z_channels=4,
                         embed_dim=4, n_embed=256)
    vsd = dm.synth_state_dict(encoder_param_spec(ecfg) + dm.decoder_param_spec(dcfg), salt=14)
    missing, unexpected = vq.load_state_dict(vsd, strict=False)
    assert not unexpected and all(k.startswith("loss") for k in missing), (missing, unexpected)
    with torch.inference_mode():
        z = seeded((2, 4, 32, 32), 46)
        out["decode_cfg4"] = dict(z=z, y=vq.decode(z))
    lat_sd = dm.synth_state_dict(dm.unet_param_spec(UnetConfig(channels=4)), salt=0)
    ref_lat = dd.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=4).eval()
    ref_lat.load_state_dict(lat_sd, strict=True)
    ldm = ld.LatentDiffusion(ref_lat, vq, latent_shape=(4, 32, 32), timesteps=1000, sampling_timesteps=6).eval()
    with patched_noise(dd, 302):
        out["ldm_cfg4_ddim6"] = dict(seed=302, B=2, S=6, y=ldm.sample(batch_size=2))
    with patched_noise(dd, 303):
        out["latent4_ddim6"] = dict(seed=303, shape=(2, 4, 32, 32), S=6, y=ldm.ddim_sample((2, 4, 32, 32)))
    del ref_lat, ldm

    # ---- config 5: text / cross-attention U-Net at 64x64 ------------------------------------------------
    tcfg = UnetConfig(text_condition=True, use_cross_attn=True)
    text_sd = dm.synth_state_dict(dm.unet_param_spec(tcfg), salt=0)
    ref_text = ddt.Unet(dim=64, dim_mults=(1, 2, 4, 8), channels=3, text_condition=True, use_cross_attn=True).eval()
    ref_text.load_state_dict(text_sd, strict=True)
    with torch.inference_mode():
        x1 = seeded((1, 3, 64, 64), 47)
        ctx1 = seeded((1, 512), 48)
        out["unet_text_full_64"] = dict(x=x1, t=tt[:1], ctx=ctx1, y=ref_text(x1, tt[:1], text_emb=ctx1))
    with tempfile.NamedTemporaryFile(suffix=".pkl") as f:  # the constructor only asserts that the file exists
        tdiff = ddt.TextConditionalDenoisingDiffusion(model=ref_text, embedding_file=f.name, image_size=64,
                                                      timesteps=1000, sampling_timesteps=4).eval()
        emb2 = seeded((2, 512), 49)
        tdiff.get_random_text_condition = lambda batch, device: (emb2[:batch], ["caption"] * batch)
        with patched_noise(ddt, 304):
            out["text64_ddim4"] = dict(seed=304, shape=(2, 3, 64, 64), S=4, ctx=emb2,
                                       y=tdiff.ddim_sample((2, 3, 64, 64)))
    del ref_text, tdiff
    # TextConditionalLatentDiffusion cannot be pinned by running it: its __init__ passes `model` positionally to the
    # keyword-only TextConditionalDenoisingDiffusion.__init__ (latent_diffusion_text_conditional.py:27 vs
    # denoising_diffusion_text_conditional.py:267) and raises TypeError as shipped.  tests/test_hip_configs.py checks our
    # class against the composition of its two pinned halves (text loop above, VQModel.decode above).
    save("configs.pt", out)


if __name__ == "__main__":
    main()
