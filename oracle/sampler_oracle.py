"""ORACLE -- test infrastructure, not product code.

CPU restatement of the reference DDPM / DDIM sampling loops with *injected*
noise, so that two implementations can be compared on identical random draws.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module.

Pinned by ``tests/golden/sampler_*.pt`` (outputs of the reference's own
``p_sample_loop`` / ``ddim_sample`` with ``torch.randn`` / ``randn_like``
redirected to the same seeded stream; see ``tests/golden/make_golden.py``).

DD = denoising-diffusion-pytorch/denoising_diffusion/ in the reference checkout.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch

Model = Callable[[torch.Tensor, torch.Tensor], torch.Tensor]


class NoiseStream:
    """Sequential N(0,1) draws from one seeded CPU generator.

    Draw 0 is x_T; draw i (i>=1) is the noise of loop iteration i-1, in the
    order the reference calls ``torch.randn`` / ``torch.randn_like``
    (DD/denoising_diffusion.py:651,643 and :676,697).
    """

    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(seed)

    def __call__(self, shape) -> torch.Tensor:
        return torch.randn(tuple(shape), generator=self.g, dtype=torch.float32)


def _tbl(sched: Dict[str, torch.Tensor], name: str, t: int) -> torch.Tensor:
    # `extract` (DD/denoising_diffusion.py:394-397) with a batch-constant t
    return sched[name][t]


def predict_start_from_noise(sched, x_t, t: int, noise):
    """DD/denoising_diffusion.py:570-574."""
    return _tbl(sched, "sqrt_recip_alphas_cumprod", t) * x_t - _tbl(sched, "sqrt_recipm1_alphas_cumprod", t) * noise


def predict_noise_from_start(sched, x_t, t: int, x0):
    """DD/denoising_diffusion.py:576-580."""
    return (_tbl(sched, "sqrt_recip_alphas_cumprod", t) * x_t - x0) / _tbl(sched, "sqrt_recipm1_alphas_cumprod", t)


def predict_start_from_v(sched, x_t, t: int, v):
    """DD/denoising_diffusion.py:588-592."""
    return _tbl(sched, "sqrt_alphas_cumprod", t) * x_t - _tbl(sched, "sqrt_one_minus_alphas_cumprod", t) * v


def _call(model: Model, x, bt, x_self_cond):
    # the model takes x_self_cond only when the loop carries one (self_condition=True)
    return model(x, bt) if x_self_cond is None else model(x, bt, x_self_cond)


def model_x_start(model: Model, sched, x, t: int, objective: str = "pred_noise", x_self_cond=None):
    """``model_predictions(...).pred_x_start`` before clipping (DD/denoising_diffusion.py:603-624)."""
    bt = torch.full((x.shape[0],), t, dtype=torch.long)
    out = _call(model, x, bt, x_self_cond)
    if objective == "pred_noise":
        return predict_start_from_noise(sched, x, t, out)
    if objective == "pred_x0":
        return out
    if objective == "pred_v":
        return predict_start_from_v(sched, x, t, out)
    raise ValueError(f"unknown objective {objective}")


def p_sample(model: Model, sched, x: torch.Tensor, t: int, noise: Optional[torch.Tensor], objective: str = "pred_noise",
             x_self_cond=None):
    """DD/denoising_diffusion.py:638-645 (+ :628-636, :594-601)."""
    x0 = model_x_start(model, sched, x, t, objective, x_self_cond).clamp(-1.0, 1.0)
    mean = _tbl(sched, "posterior_mean_coef1", t) * x0 + _tbl(sched, "posterior_mean_coef2", t) * x
    logvar = _tbl(sched, "posterior_log_variance_clipped", t)
    if t > 0:
        return mean + (0.5 * logvar).exp() * noise, x0
    return mean + (0.5 * logvar).exp() * 0.0, x0


@torch.inference_mode()
def p_sample_loop(
    model: Model,
    sched,
    shape,
    noise: Callable,
    unnormalize: bool = True,
    return_all_timesteps: bool = False,
    num_steps: Optional[int] = None,
    objective: str = "pred_noise",
    self_condition: bool = False,
):
    """DD/denoising_diffusion.py:647-664.  ``num_steps`` (oracle-only) stops
    after that many iterations, for timing a bounded sample of the loop.
    ``self_condition``: the model is called as ``model(x, t, x_self_cond)`` with the previous step's x_start (:657)."""
    T = sched["betas"].shape[0]
    img = noise(shape)
    imgs = [img]
    done = 0
    x_start = None
    for t in reversed(range(T)):
        z = noise(shape) if t > 0 else None
        sc = (x_start if x_start is not None else torch.zeros_like(img)) if self_condition else None
        img, x_start = p_sample(model, sched, img, t, z, objective, sc)
        imgs.append(img)
        done += 1
        if num_steps is not None and done >= num_steps:
            break
    ret = img if not return_all_timesteps else torch.stack(imgs, dim=1)
    return (ret + 1) * 0.5 if unnormalize else ret


def ddim_pairs(T: int, S: int) -> List[Tuple[int, int]]:
    """DD/denoising_diffusion.py:672-674."""
    times = torch.linspace(-1, T - 1, steps=S + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


@torch.inference_mode()
def ddim_sample(
    model: Model,
    sched,
    shape,
    noise: Callable,
    sampling_timesteps: int,
    eta: float = 0.0,
    unnormalize: bool = True,
    return_all_timesteps: bool = False,
    objective: str = "pred_noise",
    self_condition: bool = False,
):
    """DD/denoising_diffusion.py:666-708."""
    T = sched["betas"].shape[0]
    img = noise(shape)
    imgs = [img]
    ac = sched["alphas_cumprod"]
    x0 = None
    for t, t_next in ddim_pairs(T, sampling_timesteps):
        sc = (x0 if x0 is not None else torch.zeros_like(img)) if self_condition else None
        x0 = model_x_start(model, sched, img, t, objective, sc).clamp(-1.0, 1.0)
        eps = predict_noise_from_start(sched, img, t, x0)
        if t_next < 0:
            img = x0
            imgs.append(img)
            continue
        alpha, alpha_next = ac[t], ac[t_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        z = noise(shape)  # drawn even when sigma == 0 (DD/denoising_diffusion.py:697)
        img = x0 * alpha_next.sqrt() + c * eps + sigma * z
        imgs.append(img)
    ret = img if not return_all_timesteps else torch.stack(imgs, dim=1)
    return (ret + 1) * 0.5 if unnormalize else ret


@torch.inference_mode()
def interpolate(model: Model, sched, x1, x2, t: int, lam: float, noise: Callable):
    """DD/denoising_diffusion.py:786-803: q_sample both images at step t, mix, reverse loop from t - 1 to 0, no unnormalize."""
    a, b = sched["sqrt_alphas_cumprod"][t], sched["sqrt_one_minus_alphas_cumprod"][t]
    xt1 = a * x1 + b * noise(x1.shape)
    xt2 = a * x2 + b * noise(x2.shape)
    img = (1 - lam) * xt1 + lam * xt2
    for i in reversed(range(0, t)):
        z = noise(x1.shape) if i > 0 else None
        img, _ = p_sample(model, sched, img, i, z)
    return img


# ---- the prediction helpers with per-sample timesteps, and the guided DDIM loop ---------------------------------------
def _ext(sched, name: str, t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """``extract`` (DD/denoising_diffusion.py:394-397): gather + reshape to (B, 1, 1, 1)."""
    return sched[name][t].reshape(-1, *((1,) * (x.dim() - 1)))


def model_predictions(model: Model, sched, x, t: torch.Tensor, objective: str = "pred_noise", x_self_cond=None,
                      clip_x_start: bool = False, rederive_pred_noise: bool = False):
    """DD/denoising_diffusion.py:603-626 with a (B,) tensor of timesteps: (pred_noise, pred_x_start)."""
    out = _call(model, x, t, x_self_cond)
    clip = (lambda v: v.clamp(-1.0, 1.0)) if clip_x_start else (lambda v: v)
    rc, rm = _ext(sched, "sqrt_recip_alphas_cumprod", t, x), _ext(sched, "sqrt_recipm1_alphas_cumprod", t, x)
    if objective == "pred_noise":
        pred_noise = out
        x_start = clip(rc * x - rm * out)
        if clip_x_start and rederive_pred_noise:
            pred_noise = (rc * x - x_start) / rm
    elif objective == "pred_x0":
        x_start = clip(out)
        pred_noise = (rc * x - x_start) / rm
    else:
        x_start = clip(_ext(sched, "sqrt_alphas_cumprod", t, x) * x - _ext(sched, "sqrt_one_minus_alphas_cumprod", t, x) * out)
        pred_noise = (rc * x - x_start) / rm
    return pred_noise, x_start


def q_posterior(sched, x_start, x_t, t: torch.Tensor):
    """:594-601."""
    mean = _ext(sched, "posterior_mean_coef1", t, x_t) * x_start + _ext(sched, "posterior_mean_coef2", t, x_t) * x_t
    return mean, _ext(sched, "posterior_variance", t, x_t), _ext(sched, "posterior_log_variance_clipped", t, x_t)


def p_mean_variance(model: Model, sched, x, t: torch.Tensor, objective: str = "pred_noise", x_self_cond=None,
                    clip_denoised: bool = True):
    """:628-636."""
    _, x_start = model_predictions(model, sched, x, t, objective, x_self_cond)
    if clip_denoised:
        x_start = x_start.clamp(-1.0, 1.0)
    mean, var, logvar = q_posterior(sched, x_start, x, t)
    return mean, var, logvar, x_start


@torch.inference_mode()
def ddim_sample_guided(model: Model, sched, shape, noise: Callable, sampling_timesteps: int, eta: float = 0.0, guide=None,
                       mask=None, clip_denoised: bool = True, objective: str = "pred_noise", self_condition: bool = False):
    """:711-781 without the matplotlib figures: pred_noise is NOT re-derived from the clipped x_start, and after every update
    ``img = img * mask + q_sample(guide, time) * (1 - mask)`` (the reference diffuses the guide to ``time``, not ``time_next``)."""
    T = sched["betas"].shape[0]
    img = noise(shape)
    ac = sched["alphas_cumprod"]
    x_start = None
    for t, t_next in ddim_pairs(T, sampling_timesteps):
        bt = torch.full((shape[0],), t, dtype=torch.long)
        sc = x_start if self_condition else None
        pred_noise, x_start = model_predictions(model, sched, img, bt, objective, sc, clip_x_start=clip_denoised)
        if t_next < 0:
            img = x_start
            continue
        alpha, alpha_next = ac[t], ac[t_next]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        z = noise(shape)
        img = x_start * alpha_next.sqrt() + c * pred_noise + sigma * z
        if guide is not None:
            guide_t = sched["sqrt_alphas_cumprod"][t] * guide + sched["sqrt_one_minus_alphas_cumprod"][t] * noise(shape)
            img = img * mask + guide_t * (1 - mask)
    return (img + 1) * 0.5

