"""ORACLE -- test infrastructure, not product code.

CPU restatement of the reference's training loss and of its gradients: ``q_sample`` and ``p_losses``
(DD/denoising_diffusion.py:813-821, :823-889, with offset noise :830-834 and the immiscible noise assignment :805-817;
the hybrid KL term :880-897) over ``oracle.unet_oracle.unet_forward``, differentiated by torch autograd on the CPU.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Pinned by ``tests/golden/train.pt`` / ``train_noise.pt``: loss and parameter-gradient digests of the reference's own
``p_losses(...).backward()`` (tests/golden/make_golden_train.py, make_golden_train_noise.py); ``tests/test_oracle_golden.py`` holds this module to
them.

DD = denoising-diffusion-pytorch/denoising_diffusion/ in the reference checkout.
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import torch

from . import unet_oracle as uo

N_PROJ = 8


def directions(name: str, numel: int) -> torch.Tensor:
    """The fixed random directions the golden digests project a gradient on (make_golden_train.py)."""
    g = torch.Generator().manual_seed(zlib.crc32(("proj:" + name).encode()) & 0x7FFFFFFF)
    return torch.randn(N_PROJ, numel, generator=g, dtype=torch.float64)


def q_sample(sched, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DD/denoising_diffusion.py:813-821 (``extract`` :394-397 = gather + reshape (B,1,1,1))."""
    a = sched["sqrt_alphas_cumprod"][t].reshape(-1, 1, 1, 1)
    b = sched["sqrt_one_minus_alphas_cumprod"][t].reshape(-1, 1, 1, 1)
    return a * x_start + b * noise


def noise_assignment(x_start: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """:805-809: rows of ``noise`` assigned to the images so that the total L2 distance is smallest."""
    from scipy.optimize import linear_sum_assignment

    dist = torch.cdist(x_start.reshape(x_start.shape[0], -1), noise.reshape(noise.shape[0], -1))
    _, assign = linear_sum_assignment(dist.cpu())
    return torch.from_numpy(assign)


def target_of(sched, objective: str, x_start, t, noise):
    """:864-872; predict_v :582-586."""
    if objective == "pred_noise":
        return noise
    if objective == "pred_x0":
        return x_start
    if objective == "pred_v":
        a = sched["sqrt_alphas_cumprod"][t].reshape(-1, 1, 1, 1)
        b = sched["sqrt_one_minus_alphas_cumprod"][t].reshape(-1, 1, 1, 1)
        return a * noise - b * x_start
    raise ValueError(f"unknown objective {objective}")


def pred_x_start(sched, objective: str, x, t, out):
    """``model_predictions(...).pred_x_start`` with per-sample timesteps, unclipped (:603-626 with clip_x_start=False)."""
    e = lambda name: sched[name][t].reshape(-1, 1, 1, 1)  # noqa: E731
    if objective == "pred_noise":
        return e("sqrt_recip_alphas_cumprod") * x - e("sqrt_recipm1_alphas_cumprod") * out
    if objective == "pred_x0":
        return out
    return e("sqrt_alphas_cumprod") * x - e("sqrt_one_minus_alphas_cumprod") * out


def hybrid_kl(sched, objective: str, x, x_start, t, out):
    """The KL term of :880-897 given the output of p_mean_variance's forward pass: pred_x_start clamped to [-1, 1]
    (clip_denoised=True; ``clamp`` passes the gradient on the closed interval like ``clamp_``), q_posterior (:594-601) of it
    and of x_start, ``0.5 (plv - mlv + (exp(mlv) + (model_mean - posterior_mean)^2) / posterior_variance - 1)`` with
    ``mlv = plv``, mean per sample, ``(kl * mask).sum() / (mask.sum() + 1e-8)`` with ``mask = t > 0`` -- the division by
    ``posterior_variance[0] = 0`` in front of the mask is the reference's (a batch holding t = 0 gives NaN)."""
    e = lambda name: sched[name][t].reshape(-1, 1, 1, 1)  # noqa: E731
    x0 = pred_x_start(sched, objective, x, t, out).clamp(-1.0, 1.0)
    model_mean = e("posterior_mean_coef1") * x0 + e("posterior_mean_coef2") * x
    posterior_mean = e("posterior_mean_coef1") * x_start + e("posterior_mean_coef2") * x
    plv = e("posterior_log_variance_clipped")
    kl = 0.5 * (plv - plv + (torch.exp(plv) + (model_mean - posterior_mean) ** 2) / e("posterior_variance") - 1)
    kl = kl.reshape(kl.shape[0], -1).mean(dim=1)
    mask = (t > 0).float()
    return (kl * mask).sum() / (mask.sum() + 1e-8)


def p_losses(sd: Dict[str, torch.Tensor], cfg, sched, x_start, t, noise, objective: str = "pred_noise", self_cond=False,
             offset_noise=None, offset_noise_strength=0.0, immiscible=False, hybrid=False, kl_fwd_kw=None, **fwd_kw):
    """:823-889 with loss_weight from the schedule buffers (ones for the default ``ddpm=True``, :532-533).
    ``self_cond`` (``Unet(self_condition=True)``): the branch of :846-855 the reference takes for half of the iterations --
    a gradient-free forward pass predicts x_start, which conditions the differentiated pass.
    ``offset_noise`` (B, C) with ``offset_noise_strength`` > 0: :830-834.  ``immiscible``: q_sample mixes in
    ``noise[assign]`` (:815-817) while the target below stays the unpermuted noise, exactly as the reference has it.
    ``hybrid`` (``hybrid_loss=True``): ``+ 0.001 * kl`` with the KL term evaluated on a SECOND forward pass, as the
    reference's ``p_mean_variance`` call is one (``kl_fwd_kw``: that pass's keyword arguments when they differ, e.g. its own
    dropout masks)."""
    if offset_noise_strength > 0.0:
        noise = noise + offset_noise_strength * offset_noise.reshape(*offset_noise.shape, 1, 1)
    x = q_sample(sched, x_start, t, noise[noise_assignment(x_start, noise)] if immiscible else noise)
    if getattr(cfg, "self_condition", False) and self_cond:
        with torch.no_grad():
            det = {k: v.detach() for k, v in sd.items()}
            fwd_kw = dict(fwd_kw, x_self_cond=pred_x_start(sched, objective, x, t, uo.unet_forward(det, cfg, x, t)))
    out = uo.unet_forward(sd, cfg, x, t, **fwd_kw)
    loss = torch.nn.functional.mse_loss(out, target_of(sched, objective, x_start, t, noise), reduction="none")
    loss = loss.reshape(loss.shape[0], -1).mean(dim=1) * sched["loss_weight"][t]
    if hybrid:
        out2 = uo.unet_forward(sd, cfg, x, t, **(fwd_kw if kl_fwd_kw is None else dict(fwd_kw, **kl_fwd_kw)))
        loss = loss + 0.001 * hybrid_kl(sched, objective, x, x_start, t, out2)
    return loss.mean()


def loss_and_grads(sd, cfg, sched, x_start, t, noise, objective: str = "pred_noise",
                   **fwd_kw) -> Tuple[float, Dict[str, torch.Tensor]]:
    """(loss, {parameter name: gradient}) of one ``p_losses`` call."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = p_losses(leaf, cfg, sched, x_start, t, noise, objective, **fwd_kw)
    loss.backward()
    return float(loss.detach()), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
