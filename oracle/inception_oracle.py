"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch fp32 ops) of InceptionV3 as the reference's evaluators call it:

* ``fid_features``: ``pytorch_fid.inception.InceptionV3([3])(x)[0]`` as used by
  denoising_diffusion/fid_evaluation.py:41-51 -- bilinear resize to 299x299 (align_corners=False), ``2x - 1``, the
  torchvision graph with pytorch_fid's three deviations (``count_include_pad=False`` in the average pools of
  Mixed_5b-6e and Mixed_7b; a MAX pool in Mixed_7c's pool branch), global average -> (B, 2048, 1, 1).
* ``is_logits``: ``torchvision.models.inception_v3(weights=..., aux_logits=True).eval()(x)`` as used by
  denoising_diffusion/inception_score_evaluation.py:70-92 -- the caller's ImageNet normalisation, then
  ``transform_input`` (set by torchvision whenever pretrained weights are loaded), the plain graph, dropout = id, fc.
* ``frechet_distance``: ``pytorch_fid.fid_score.calculate_frechet_distance``.

Both libraries are pip dependencies absent from the reference tree and from this image, and their weights cannot be
fetched: **parity unpinned**.  The architecture is restated from the published sources (torchvision 0.17
models/inception.py; pytorch-fid 0.3.0 inception.py / fid_score.py) independently of
``diffusion_models_amd/inception.py``; only the layer table (names, shapes) is shared (inception_spec.py).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _bc(sd: SD, p: str, x, stride=1, padding=0):
    """BasicConv2d: conv (no bias) -> BatchNorm2d(eps=0.001), eval -> ReLU."""
    x = F.conv2d(x, sd[p + ".conv.weight"], None, stride=stride, padding=padding)
    x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"], sd[p + ".bn.bias"],
                     training=False, eps=0.001)
    return F.relu(x)


def _pool_branch(x, fid: bool):
    return F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=not fid)


def _a(sd, p, x, fid):
    b1 = _bc(sd, p + ".branch1x1", x)
    b5 = _bc(sd, p + ".branch5x5_2", _bc(sd, p + ".branch5x5_1", x), padding=2)
    b3 = _bc(sd, p + ".branch3x3dbl_1", x)
    b3 = _bc(sd, p + ".branch3x3dbl_3", _bc(sd, p + ".branch3x3dbl_2", b3, padding=1), padding=1)
    bp = _bc(sd, p + ".branch_pool", _pool_branch(x, fid))
    return torch.cat([b1, b5, b3, bp], 1)


def _b(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3", x, stride=2)
    bd = _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), padding=1)
    bd = _bc(sd, p + ".branch3x3dbl_3", bd, stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def _c(sd, p, x, fid):
    b1 = _bc(sd, p + ".branch1x1", x)
    b7 = _bc(sd, p + ".branch7x7_1", x)
    b7 = _bc(sd, p + ".branch7x7_2", b7, padding=(0, 3))
    b7 = _bc(sd, p + ".branch7x7_3", b7, padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_1", x)
    bd = _bc(sd, p + ".branch7x7dbl_2", bd, padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_3", bd, padding=(0, 3))
    bd = _bc(sd, p + ".branch7x7dbl_4", bd, padding=(3, 0))
    bd = _bc(sd, p + ".branch7x7dbl_5", bd, padding=(0, 3))
    bp = _bc(sd, p + ".branch_pool", _pool_branch(x, fid))
    return torch.cat([b1, b7, bd, bp], 1)


def _d(sd, p, x):
    b3 = _bc(sd, p + ".branch3x3_2", _bc(sd, p + ".branch3x3_1", x), stride=2)
    b7 = _bc(sd, p + ".branch7x7x3_1", x)
    b7 = _bc(sd, p + ".branch7x7x3_2", b7, padding=(0, 3))
    b7 = _bc(sd, p + ".branch7x7x3_3", b7, padding=(3, 0))
    b7 = _bc(sd, p + ".branch7x7x3_4", b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, kernel_size=3, stride=2)], 1)


def _e(sd, p, x, pool: str):
    b1 = _bc(sd, p + ".branch1x1", x)
    b3 = _bc(sd, p + ".branch3x3_1", x)
    b3 = torch.cat([_bc(sd, p + ".branch3x3_2a", b3, padding=(0, 1)), _bc(sd, p + ".branch3x3_2b", b3, padding=(1, 0))], 1)
    bd = _bc(sd, p + ".branch3x3dbl_2", _bc(sd, p + ".branch3x3dbl_1", x), padding=1)
    bd = torch.cat([_bc(sd, p + ".branch3x3dbl_3a", bd, padding=(0, 1)), _bc(sd, p + ".branch3x3dbl_3b", bd, padding=(1, 0))], 1)
    if pool == "max":  # pytorch_fid's FIDInceptionE_2
        bp = F.max_pool2d(x, kernel_size=3, stride=1, padding=1)
    else:
        bp = F.avg_pool2d(x, kernel_size=3, stride=1, padding=1, count_include_pad=(pool == "avg"))
    bp = _bc(sd, p + ".branch_pool", bp)
    return torch.cat([b1, b3, bd, bp], 1)


def trunk(sd: SD, x: torch.Tensor, fid: bool) -> torch.Tensor:
    """x: (B, 3, 299, 299) already normalised -> (B, 2048, 8, 8)."""
    x = _bc(sd, "Conv2d_1a_3x3", x, stride=2)
    x = _bc(sd, "Conv2d_2a_3x3", x)
    x = _bc(sd, "Conv2d_2b_3x3", x, padding=1)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    x = _bc(sd, "Conv2d_3b_1x1", x)
    x = _bc(sd, "Conv2d_4a_3x3", x)
    x = F.max_pool2d(x, kernel_size=3, stride=2)
    x = _a(sd, "Mixed_5b", x, fid)
    x = _a(sd, "Mixed_5c", x, fid)
    x = _a(sd, "Mixed_5d", x, fid)
    x = _b(sd, "Mixed_6a", x)
    x = _c(sd, "Mixed_6b", x, fid)
    x = _c(sd, "Mixed_6c", x, fid)
    x = _c(sd, "Mixed_6d", x, fid)
    x = _c(sd, "Mixed_6e", x, fid)
    x = _d(sd, "Mixed_7a", x)
    x = _e(sd, "Mixed_7b", x, "avg_valid" if fid else "avg")
    x = _e(sd, "Mixed_7c", x, "max" if fid else "avg")
    return x


@torch.inference_mode()
def fid_features(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """pytorch_fid InceptionV3([3], resize_input=True, normalize_input=True): x in [0, 1] -> (B, 2048, 1, 1)."""
    x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    x = 2 * x - 1
    return F.adaptive_avg_pool2d(trunk(sd, x, fid=True), (1, 1))


@torch.inference_mode()
def is_logits(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """inception_score_evaluation.py:77-92: x in [-1, 1] or [0, 1] -> logits (B, 1000)."""
    if x.min() < 0:
        x = (x + 1) / 2.0
    if x.shape[-2:] != (299, 299):
        x = F.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    x = (x - mean) / std
    # torchvision Inception3._transform_input (transform_input=True with pretrained weights)
    x0 = x[:, 0:1] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
    x1 = x[:, 1:2] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
    x2 = x[:, 2:3] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
    x = torch.cat((x0, x1, x2), 1)
    f = F.adaptive_avg_pool2d(trunk(sd, x, fid=False), (1, 1)).flatten(1)
    return F.linear(f, sd["fc.weight"], sd["fc.bias"])


def inception_score(probs: torch.Tensor) -> float:
    """inception_score_evaluation.py:94-101."""
    p_y = probs.mean(dim=0)
    eps = 1e-10
    kl = (probs * (torch.log(probs + eps) - torch.log(p_y + eps))).sum(dim=1)
    return float(np.exp(kl.mean().item()))


def frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6) -> float:
    """pytorch_fid.fid_score.calculate_frechet_distance."""
    from scipy import linalg

    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))
