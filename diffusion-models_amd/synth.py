"""Name-seeded synthetic weights.

There is no network and the reference ships no checkpoints, so every bench,
smoke and parity run uses random-init weights of the right architecture.  The
143 MB of U-Net weights are never stored: each tensor is regenerated from a
seed derived from its *name*, so the same state dict appears here, in the
golden-vector generator (``tests/golden/make_golden.py``) and on the GPU box.

Scales follow PyTorch's default initialisers for the layer types the reference
uses (Conv2d / Linear: U(+-1/sqrt(fan_in)); ``mem_kv``: N(0,1),
denoising_diffusion.py:165,211); norm gains are drawn around 1 rather than
exactly 1 so that a kernel that forgets a gain fails its parity test.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Tuple

import torch


def _gen(name: str, salt: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (salt * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(name: str, shape: Tuple[int, ...], salt: int = 0) -> torch.Tensor:
    g = _gen(name, salt)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "g":  # RMSNorm / RMSNorm1D gain
        return 1.0 + 0.25 * torch.randn(shape, generator=g)
    if leaf in ("mem_kv", "weights"):  # `weights`: RandomOrLearnedSinusoidalPosEmb, N(0, 1) (denoising_diffusion.py:94)
        return torch.randn(shape, generator=g)
    if leaf in ("weight", "bias") and (".norm" in name or "norm_out" in name) and len(shape) == 1:
        # GroupNorm affine (VAE decoder)
        if leaf == "weight":
            return 1.0 + 0.25 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)
    if ".bn." in name:  # BatchNorm of InceptionV3's BasicConv2d (eval mode: affine + running statistics)
        if leaf == "weight":
            return 1.0 + 0.2 * (torch.rand(shape, generator=g) * 2 - 1)
        if leaf == "running_var":
            return 0.5 + torch.rand(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)  # bias, running_mean
    if name.endswith(".conv.weight") and name.startswith(("Conv2d_", "Mixed_")):
        # InceptionV3 conv in front of a ReLU: He-uniform keeps activations O(1) through 90 layers
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        return (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(6.0 / fan_in)
    if leaf == "weight":
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        bound = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if leaf == "bias":
        # the bound PyTorch would use needs fan_in of the sibling weight; a fixed
        # modest scale keeps the name->tensor map self-contained
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    raise ValueError(f"no synthetic rule for {name}")


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], salt: int = 0) -> Dict[str, torch.Tensor]:
    return {name: synth_tensor(name, tuple(shape), salt).to(torch.float32).contiguous() for name, shape in spec}
