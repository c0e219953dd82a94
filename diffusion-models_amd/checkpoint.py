"""Checkpoint ingestion: the on-disk formats next to the sampling path.

* ``Trainer.save`` writes ``model-{milestone}.pt`` = ``{'step', 'model', 'opt', 'ema', 'scaler', 'version'}``
  (denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion.py:1100-1113).  The sampling scripts use
  ``data['ema']``, the state dict of an ``ema_pytorch.EMA`` wrapper whose averaged copy lives under the prefix
  ``ema_model.`` (denoising-diffusion-pytorch/sampling.py:157-159); ``data['model']`` holds the raw weights.
* The VAE is a Lightning ``.ckpt`` whose ``state_dict`` (or the file itself) holds ``decoder.*`` /
  ``post_quant_conv.*`` (latent-diffusion/train/train_ldm.py:36-43).

Files are opened with ``weights_only=True`` only (nothing from the file is executed).
"""
from __future__ import annotations

from typing import Dict

import torch

EMA_PREFIX = "ema_model."


def diffusion_state_dict_from_checkpoint(data: Dict, prefer_ema: bool = True) -> Dict[str, torch.Tensor]:
    """``DenoisingDiffusion.state_dict()``-shaped dict (13 buffers + ``model.*``) out of a Trainer checkpoint."""
    if prefer_ema and "ema" in data:
        ema = data["ema"]
        sd = {k[len(EMA_PREFIX):]: v for k, v in ema.items() if k.startswith(EMA_PREFIX)}
        if sd:
            return sd
        raise KeyError(f"checkpoint['ema'] has no '{EMA_PREFIX}*' entries")
    if "model" in data:
        return dict(data["model"])
    raise KeyError("checkpoint has neither 'ema' nor 'model'")


def load_trainer_checkpoint(path: str, prefer_ema: bool = True) -> Dict[str, torch.Tensor]:
    data = torch.load(path, map_location="cpu", weights_only=True)
    return diffusion_state_dict_from_checkpoint(data, prefer_ema)


def vae_state_dict_from_checkpoint(data: Dict) -> Dict[str, torch.Tensor]:
    """Lightning ``.ckpt`` (``{'state_dict': ...}``) or a bare state dict -> the entries ``VQDecoder`` / ``VQEncoder`` /
    ``VQModel`` need (the loss network and EMA shadows of the checkpoint are dropped)."""
    sd = data["state_dict"] if "state_dict" in data else data
    keep = ("decoder.", "post_quant_conv.", "encoder.", "quant_conv.", "quantize.embedding.")
    return {k: v for k, v in sd.items() if k.startswith(keep)}


def load_vae_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    return vae_state_dict_from_checkpoint(torch.load(path, map_location="cpu", weights_only=True))
