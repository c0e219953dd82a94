#!/usr/bin/env python3
"""Goldens for the prediction helpers and the guided DDIM loop of the reference's ``DenoisingDiffusion`` (build container only):
``python tests/golden/make_golden_guided.py``  ->  ``tests/golden/guided.pt``.

* ``model_predictions`` (DD/denoising_diffusion.py:603-626, with and without clipping / re-derived noise), ``p_mean_variance``
  (:628-636), ``q_posterior`` (:594-601), ``predict_v`` / ``predict_start_from_v`` / ``predict_noise_from_start`` with a
  batch of DIFFERENT timesteps, for the three objectives;
* ``ddim_sample_guided`` (:711-781) with a guide image and a mask, ``torch.randn`` / ``randn_like`` redirected to one seeded
  stream; matplotlib runs on the non-interactive Agg backend (``plt.show()`` does nothing), its figures are closed.

Only DATA is written."""
from __future__ import annotations

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

os.environ.setdefault("MPLBACKEND", "Agg")

from make_golden import import_reference, patched_noise, save, seeded  # noqa: E402

import diffusion_models_amd as dm  # noqa: E402
from diffusion_models_amd.spec import UnetConfig  # noqa: E402


def main():
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt

    torch.manual_seed(0)
    torch.set_num_threads(8)
    dd, _, _ = import_reference()
    cfg = UnetConfig(dim=64, dim_mults=(1, 2), channels=3)
    sd = dm.synth_state_dict(dm.unet_param_spec(cfg), salt=31)
    unet = dd.Unet(dim=64, dim_mults=(1, 2), channels=3).eval()
    unet.load_state_dict(sd, strict=True)
    out = {}
    x = seeded((4, 3, 16, 16), 81)
    t = torch.tensor([0, 17, 500, 999], dtype=torch.long)
    for obj in ("pred_noise", "pred_x0", "pred_v"):
        d = dd.DenoisingDiffusion(unet, image_size=16, timesteps=1000, objective=obj).eval()
        with torch.inference_mode():
            a = d.model_predictions(x, t)
            b = d.model_predictions(x, t, clip_x_start=True, rederive_pred_noise=True)
            mean, var, logvar, xs = d.p_mean_variance(x, t)
            other = seeded((4, 3, 16, 16), 82)
            out[f"pred_{obj}"] = dict(x=x, t=t, pred_noise=a.pred_noise, pred_x_start=a.pred_x_start,
                                      pred_noise_clip=b.pred_noise, pred_x_start_clip=b.pred_x_start, mean=mean, var=var,
                                      logvar=logvar, x_start=xs, other=other, predict_v=d.predict_v(x, t, other),
                                      predict_start_from_v=d.predict_start_from_v(x, t, other),
                                      predict_noise_from_start=d.predict_noise_from_start(x, t, other),
                                      predict_start_from_noise=d.predict_start_from_noise(x, t, other))
    # guided DDIM
    g = torch.Generator().manual_seed(5)
    guide = torch.rand((2, 3, 16, 16), generator=g) * 2 - 1
    mask = torch.zeros((1, 1, 16, 16))
    mask[..., :, 8:] = 1.0  # keep the right half of the sample, take the left half from the guide
    d = dd.DenoisingDiffusion(unet, image_size=16, timesteps=1000, sampling_timesteps=4, ddim_sampling_eta=0.5).eval()
    with patched_noise(dd, 360):
        y = d.ddim_sample_guided((2, 3, 16, 16), guide=guide, mask=mask)
    plt.close("all")
    with patched_noise(dd, 361):
        y0 = d.ddim_sample_guided((2, 3, 16, 16))
    out["guided"] = dict(seed=360, shape=(2, 3, 16, 16), S=4, eta=0.5, guide=guide, mask=mask, y=y, seed_noguide=361, y_noguide=y0)
    save("guided.pt", out)


if __name__ == "__main__":
    main()
