#!/usr/bin/env python3
"""Goldens for the two noise options of the reference's training loss, from the REFERENCE itself (build container only):
``python tests/golden/make_golden_train_noise.py``  ->  ``tests/golden/train_noise.pt``.

* offset noise (DD/denoising_diffusion.py:830-834): ``p_losses(..., offset_noise_strength=0.1)`` with the ``torch.randn``
  draw of the (B, C) offsets redirected to a stored tensor;
* immiscible diffusion (:805-817): ``DenoisingDiffusion(immiscible=True)``: ``q_sample`` re-assigns the noise rows
  (``torch.cdist`` + scipy's ``linear_sum_assignment``) while ``p_losses`` keeps the unpermuted noise as the target.

Loss, q_sample output, the assignment and the gradient digests of make_golden_train.py.  Only DATA is written."""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from make_golden import import_reference, save, seeded  # noqa: E402
from make_golden_train import digest  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd.spec import UnetConfig  # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dd, _, _ = import_reference()
    cfg = UnetConfig(dim=32, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=41)
    B, side, T = 8, 16, 1000
    out = {}

    def model(**kw):
        unet = dd.Unet(dim=32, dim_mults=(1, 2), channels=3)
        unet.load_state_dict(sd, strict=True)
        return unet, dd.DenoisingDiffusion(unet, image_size=side, timesteps=T, **kw).train()

    img = torch.rand((B, 3, side, side), generator=torch.Generator().manual_seed(600))
    t = torch.randint(0, T, (B,), generator=torch.Generator().manual_seed(601))
    noise = seeded((B, 3, side, side), 602)

    # ---- offset noise
    unet, diff = model()
    offs = seeded((B, 3), 603)
    real = dd.torch

    class _T:
        def __getattr__(_, k):
            if k == "randn":
                return lambda shape, device=None, **kw: offs.clone()
            return getattr(real, k)

    dd.torch = _T()
    try:
        loss = diff.p_losses(diff.normalize(img), t, noise=noise.clone(), offset_noise_strength=0.1)
    finally:
        dd.torch = real
    loss.backward()
    out["offset"] = dict(img=img, t=t, noise=noise, offset=offs, strength=0.1, loss=float(loss), T=T,
                         grads={k: digest(k, p.grad) for k, p in unet.named_parameters()})

    # ---- immiscible
    unet, diff = model(immiscible=True)
    x_start = diff.normalize(img)
    loss = diff.p_losses(x_start, t, noise=noise.clone())
    loss.backward()
    with torch.no_grad():
        assign = diff.noise_assignment(x_start, noise)
        xq = diff.q_sample(x_start, t, noise)
    out["immiscible"] = dict(img=img, t=t, noise=noise, assign=assign.long(), x_noisy=xq, loss=float(loss), T=T,
                             grads={k: digest(k, p.grad) for k, p in unet.named_parameters()})
    print("assign", assign.tolist())
    save("train_noise.pt", out)


if __name__ == "__main__":
    main()
