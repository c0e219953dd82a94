"""ORACLE -- test infrastructure, not product code.

CPU restatement of the LDM decode step: ``VQModel.decode`` =
``post_quant_conv`` (1x1) -> ``Decoder`` (GroupNorm(32, eps 1e-6) -> swish ->
conv3x3 residual blocks, single-head n x n attention, nearest x2 + conv3x3).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py`` may import it.

Pinned by ``tests/golden/vae_*.pt`` (generated from the reference's
``ldm/modules/diffusionmodules/model.py`` which imports without stubs).
``VQModel`` itself needs pytorch_lightning + taming (absent); its ``decode`` is
the two calls restated in :func:`vq_decode` (autoencoder.py:113-116).

LD = latent-diffusion/ldm/ in the reference checkout.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _gn(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    # LD/modules/diffusionmodules/model.py:55-56
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _swish(x: torch.Tensor) -> torch.Tensor:
    # LD/modules/diffusionmodules/model.py:50-52
    return x * torch.sigmoid(x)


def vae_resnet_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """LD/modules/diffusionmodules/model.py:138-158 with temb=None, dropout off."""
    h = F.conv2d(_swish(_gn(sd, p + ".norm1", x)), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.conv2d(_swish(_gn(sd, p + ".norm2", h)), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".nin_shortcut.weight"], sd[p + ".nin_shortcut.bias"])
    return x + h


def vae_attn_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """LD/modules/diffusionmodules/model.py:195-219."""
    b, c, h, w = x.shape
    hn = _gn(sd, p + ".norm", x)
    q = F.conv2d(hn, sd[p + ".q.weight"], sd[p + ".q.bias"]).reshape(b, c, h * w)
    k = F.conv2d(hn, sd[p + ".k.weight"], sd[p + ".k.bias"]).reshape(b, c, h * w)
    v = F.conv2d(hn, sd[p + ".v.weight"], sd[p + ".v.bias"]).reshape(b, c, h * w)
    w_ = torch.bmm(q.permute(0, 2, 1), k) * (int(c) ** (-0.5))  # (b, i, j)
    w_ = F.softmax(w_, dim=2)
    out = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    out = F.conv2d(out, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return x + out


def decoder_forward(sd: SD, cfg, z: torch.Tensor, prefix: str = "decoder") -> torch.Tensor:
    """LD/modules/diffusionmodules/model.py:552-585 (give_pre_end=False, tanh_out=False)."""
    p = prefix
    h = F.conv2d(z, sd[p + ".conv_in.weight"], sd[p + ".conv_in.bias"], padding=1)
    h = vae_resnet_block(sd, p + ".mid.block_1", h)
    h = vae_attn_block(sd, p + ".mid.attn_1", h)
    h = vae_resnet_block(sd, p + ".mid.block_2", h)
    for lvl in reversed(range(cfg.num_resolutions)):
        for b in range(cfg.num_res_blocks + 1):
            h = vae_resnet_block(sd, f"{p}.up.{lvl}.block.{b}", h)
            if f"{p}.up.{lvl}.attn.{b}.norm.weight" in sd:
                h = vae_attn_block(sd, f"{p}.up.{lvl}.attn.{b}", h)
        if lvl != 0:
            h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
            h = F.conv2d(h, sd[f"{p}.up.{lvl}.upsample.conv.weight"], sd[f"{p}.up.{lvl}.upsample.conv.bias"], padding=1)
    h = _swish(_gn(sd, p + ".norm_out", h))
    return F.conv2d(h, sd[p + ".conv_out.weight"], sd[p + ".conv_out.bias"], padding=1)


def vq_decode(sd: SD, cfg, quant: torch.Tensor) -> torch.Tensor:
    """LD/models/autoencoder.py:113-116."""
    q = F.conv2d(quant, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    return decoder_forward(sd, cfg, q)


def encoder_forward(sd: SD, cfg, x: torch.Tensor, prefix: str = "encoder") -> torch.Tensor:
    """LD/modules/diffusionmodules/model.py:451-476 (Encoder.forward, temb=None); Downsample :89-93 pads (0,1,0,1)
    and convolves 3x3 with stride 2."""
    p = prefix
    h = F.conv2d(x, sd[p + ".conv_in.weight"], sd[p + ".conv_in.bias"], padding=1)
    for lvl in range(cfg.num_resolutions):
        for b in range(cfg.num_res_blocks):
            h = vae_resnet_block(sd, f"{p}.down.{lvl}.block.{b}", h)
            if f"{p}.down.{lvl}.attn.{b}.norm.weight" in sd:
                h = vae_attn_block(sd, f"{p}.down.{lvl}.attn.{b}", h)
        if lvl != cfg.num_resolutions - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"{p}.down.{lvl}.downsample.conv.weight"], sd[f"{p}.down.{lvl}.downsample.conv.bias"],
                         stride=2, padding=0)
    h = vae_resnet_block(sd, p + ".mid.block_1", h)
    h = vae_attn_block(sd, p + ".mid.attn_1", h)
    h = vae_resnet_block(sd, p + ".mid.block_2", h)
    h = _swish(_gn(sd, p + ".norm_out", h))
    return F.conv2d(h, sd[p + ".conv_out.weight"], sd[p + ".conv_out.bias"], padding=1)


def vq_encode_to_prequant(sd: SD, cfg, x: torch.Tensor) -> torch.Tensor:
    """LD/models/autoencoder.py:108-111."""
    return F.conv2d(encoder_forward(sd, cfg, x), sd["quant_conv.weight"], sd["quant_conv.bias"])


def vector_quantize(sd: SD, z: torch.Tensor):
    """PARITY UNPINNED: the quantizer is ``taming.modules.vqvae.quantize.VectorQuantizer2`` (taming-transformers, a pip
    dependency imported at LD/models/autoencoder.py:11 and absent from the reference tree and from this image), restated
    from its published source (forward, legacy / remap off): distances ``|z|^2 + |e|^2 - 2 z.e`` over the flattened
    (b h w, c) rows, ``argmin``, embedding lookup, straight-through ``z + (z_q - z).detach()``, back to (b c h w).
    Returns (z_q, indices)."""
    e = sd["quantize.embedding.weight"]
    zf = z.permute(0, 2, 3, 1).contiguous()
    flat = zf.view(-1, e.shape[1])
    d = (flat ** 2).sum(dim=1, keepdim=True) + (e ** 2).sum(dim=1) - 2 * torch.einsum("bd,dn->bn", flat, e.t())
    idx = torch.argmin(d, dim=1)
    zq = e[idx].view(zf.shape)
    zq = zf + (zq - zf)
    return zq.permute(0, 3, 1, 2).contiguous(), idx


def vq_encode(sd: SD, cfg, x: torch.Tensor):
    """LD/models/autoencoder.py:102-106: (quant, indices)."""
    return vector_quantize(sd, vq_encode_to_prequant(sd, cfg, x))

