"""Checkpoint ingestion on CPU: a Trainer-style file written by this test (the reference ships none)."""
import os

import pytest
import torch

import diffusion_models_amd as dm
from diffusion_models_amd import checkpoint as ck
from diffusion_models_amd.spec import DecoderConfig, UnetConfig, SCHEDULE_BUFFERS


def _diffusion_sd():
    cfg = UnetConfig(dim=32, dim_mults=(1, 2))
    sd = {f"model.{k}": v for k, v in dm.synth_state_dict(dm.unet_param_spec(cfg), salt=9).items()}
    sd.update(dm.make_schedule(1000, "linear"))
    return sd


def test_trainer_checkpoint_roundtrip(tmp_path):
    sd = _diffusion_sd()
    ema = {"ema_model." + k: v for k, v in sd.items()}
    ema.update({"online_model." + k: v + 1 for k, v in sd.items()})
    ema["initted"] = torch.tensor(True)
    ema["step"] = torch.tensor(7)
    path = os.path.join(tmp_path, "model-3.pt")
    torch.save({"step": 1500, "model": {k: v + 1 for k, v in sd.items()}, "opt": {}, "ema": ema, "scaler": None,
                "version": "x"}, path)
    got = ck.load_trainer_checkpoint(path)
    assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    assert set(SCHEDULE_BUFFERS) <= set(got)
    raw = ck.load_trainer_checkpoint(path, prefer_ema=False)
    assert torch.equal(raw["model.init_conv.bias"], sd["model.init_conv.bias"] + 1)
    with pytest.raises(KeyError):
        ck.diffusion_state_dict_from_checkpoint({"opt": {}})


def test_vae_checkpoint_filter(tmp_path):
    cfg = DecoderConfig()
    sd = dm.synth_state_dict(dm.decoder_param_spec(cfg), salt=4)
    full = dict(sd)
    full["encoder.conv_in.weight"] = torch.zeros(1)
    full["quantize.embedding.weight"] = torch.zeros(1)
    full["loss.discriminator.main.0.weight"] = torch.zeros(1)  # training-only parts of the Lightning checkpoint
    full["model_ema.decay"] = torch.zeros(1)
    path = os.path.join(tmp_path, "vae.ckpt")
    torch.save({"state_dict": full, "epoch": 3}, path)
    got = ck.load_vae_checkpoint(path)
    assert set(got) == set(sd) | {"encoder.conv_in.weight", "quantize.embedding.weight"}
