"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch fp32 ops, functional style over a flat state
dict) of the reference denoising U-Net forward.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``diffusion-models_amd``) never does.

Pinning: every function below is checked against golden vectors produced by
importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.pt``; see
``tests/test_oracle_golden.py``).  The reference has no tests or fixtures of
its own, so those generated vectors are the pin.

Citations: DD = denoising-diffusion-pytorch/denoising_diffusion/ in the
reference checkout.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --- leaf ops ---------------------------------------------------------------

def rms_norm(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """DD/denoising_diffusion.py:60-67.  L2-normalise over channels (eps 1e-12
    on the norm, as F.normalize), times g, times sqrt(C)."""
    c = x.shape[1]
    n = x.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / n * g * (c ** 0.5)


def sinusoidal_pos_emb(t: torch.Tensor, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """DD/denoising_diffusion.py:77-84."""
    half = dim // 2
    k = math.log(theta) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=t.device) * -k)
    ang = t[:, None] * freqs[None, :]
    return torch.cat((ang.sin(), ang.cos()), dim=-1)


def learned_sinusoidal_pos_emb(t: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """RandomOrLearnedSinusoidalPosEmb.forward (DD/denoising_diffusion.py:96-101): cat(x, sin(x w 2 pi), cos(x w 2 pi))."""
    x = t.to(weights.dtype)[:, None]
    freqs = x * weights[None, :] * 2 * math.pi
    return torch.cat((x, freqs.sin(), freqs.cos()), dim=-1)


def time_mlp(sd: SD, p: str, t: torch.Tensor, dim: int, theta: float) -> torch.Tensor:
    """DD/denoising_diffusion.py:280-285: sinusoid -> Linear -> GELU(erf) -> Linear."""
    if p + "time_mlp.0.weights" in sd:  # random / learned sinusoidal embedding (:271-273)
        e = learned_sinusoidal_pos_emb(t, sd[p + "time_mlp.0.weights"])
    else:
        e = sinusoidal_pos_emb(t, dim, theta)
    e = F.linear(e, sd[p + "time_mlp.1.weight"], sd[p + "time_mlp.1.bias"])
    e = F.gelu(e)
    return F.linear(e, sd[p + "time_mlp.3.weight"], sd[p + "time_mlp.3.bias"])


def block(sd: SD, p: str, x: torch.Tensor, scale_shift=None, drop=None) -> torch.Tensor:
    """DD/denoising_diffusion.py:113-122: conv3x3 -> RMSNorm -> x*(scale+1)+shift -> SiLU -> dropout.
    ``drop`` (training mode only): the dropout factor per element, 0 or 1 / (1 - p), injected so that two
    implementations can be compared on the same mask (nn.Dropout draws it from torch's generator)."""
    x = F.conv2d(x, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=1)
    x = rms_norm(x, sd[p + ".norm.g"])
    if scale_shift is not None:
        scale, shift = scale_shift
        x = x * (scale + 1) + shift
    x = F.silu(x)
    return x if drop is None else x * drop


def resnet_block(sd: SD, p: str, x: torch.Tensor, t_emb: Optional[torch.Tensor], drops=None) -> torch.Tensor:
    """DD/denoising_diffusion.py:136-148.  ``drops``: iterator of dropout factors, one per Block in forward order."""
    ss = None
    if t_emb is not None and (p + ".mlp.1.weight") in sd:
        e = F.linear(F.silu(t_emb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        e = e[:, :, None, None]
        ss = e.chunk(2, dim=1)
    h = block(sd, p + ".block1", x, ss, next(drops) if drops is not None else None)
    h = block(sd, p + ".block2", h, None, next(drops) if drops is not None else None)
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def linear_attention(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int) -> torch.Tensor:
    """DD/denoising_diffusion.py:173-193."""
    b, c, h, w = x.shape
    n = h * w
    xn = rms_norm(x, sd[p + ".norm.g"])
    qkv = F.conv2d(xn, sd[p + ".to_qkv.weight"])
    q, k, v = (t.reshape(b, heads, dim_head, n) for t in qkv.chunk(3, dim=1))
    mem = sd[p + ".mem_kv"]  # (2, heads, dim_head, m)
    mk = mem[0].unsqueeze(0).expand(b, -1, -1, -1)
    mv = mem[1].unsqueeze(0).expand(b, -1, -1, -1)
    k = torch.cat((mk, k), dim=-1)
    v = torch.cat((mv, v), dim=-1)
    q = q.softmax(dim=-2) * (dim_head ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q)
    out = out.reshape(b, heads * dim_head, h, w)
    out = F.conv2d(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return rms_norm(out, sd[p + ".to_out.1.g"])


def full_attention(sd: SD, p: str, x: torch.Tensor, heads: int, dim_head: int) -> torch.Tensor:
    """DD/denoising_diffusion.py:215-229 with the non-flash branch of DD/attend.py:109-124."""
    b, c, h, w = x.shape
    n = h * w
    xn = rms_norm(x, sd[p + ".norm.g"])
    qkv = F.conv2d(xn, sd[p + ".to_qkv.weight"])
    q, k, v = (t.reshape(b, heads, dim_head, n).transpose(-1, -2) for t in qkv.chunk(3, dim=1))
    mem = sd[p + ".mem_kv"]  # (2, heads, m, dim_head)
    mk = mem[0].unsqueeze(0).expand(b, -1, -1, -1)
    mv = mem[1].unsqueeze(0).expand(b, -1, -1, -1)
    k = torch.cat((mk, k), dim=-2)
    v = torch.cat((mv, v), dim=-2)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (dim_head ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.transpose(-1, -2).reshape(b, heads * dim_head, h, w)
    return F.conv2d(out, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])


def downsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """DD/denoising_diffusion.py:54-58: 'b c (h 2) (w 2) -> b (c 2 2) h w' then conv1x1."""
    b, c, hh, ww = x.shape
    x = x.reshape(b, c, hh // 2, 2, ww // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(b, c * 4, hh // 2, ww // 2)
    return F.conv2d(x, sd[p + ".1.weight"], sd[p + ".1.bias"])


def upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """DD/denoising_diffusion.py:48-52: nearest x2 then conv3x3."""
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return F.conv2d(x, sd[p + ".1.weight"], sd[p + ".1.bias"], padding=1)


def rms_norm_1d(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """DD/denoising_diffusion_text_conditional.py:27-36 (normalise the last dim)."""
    n = x.pow(2).sum(dim=-1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / n * g * (x.shape[-1] ** 0.5)


def cross_attention(sd: SD, p: str, x: torch.Tensor, context: torch.Tensor, heads: int = 4) -> torch.Tensor:
    """DD/denoising_diffusion_text_conditional.py:54-78.  x (b,n,dim); context (b,m,ctx) or (b,ctx)."""
    if context.ndim == 2:
        context = context.unsqueeze(1)
    b, n, _ = x.shape
    m = context.shape[1]
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(context, sd[p + ".to_k.weight"])
    v = F.linear(context, sd[p + ".to_v.weight"])
    d = q.shape[-1] // heads
    q = q.reshape(b, n, heads, d).transpose(1, 2)
    k = k.reshape(b, m, heads, d).transpose(1, 2)
    v = v.reshape(b, m, heads, d).transpose(1, 2)
    attn = (torch.einsum("bhnd,bhmd->bhnm", q, k) * (d ** -0.5)).softmax(dim=-1)
    out = torch.einsum("bhnm,bhmd->bhnd", attn, v).transpose(1, 2).reshape(b, n, heads * d)
    out = F.linear(out, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return rms_norm_1d(out, sd[p + ".to_out.1.g"])


# --- whole network ------------------------------------------------------------

def _apply_cross(sd: SD, p: str, x: torch.Tensor, text_emb: torch.Tensor) -> torch.Tensor:
    # DD/denoising_diffusion_text_conditional.py:173-177 -- the result REPLACES x
    b, c, h, w = x.shape
    flat = x.reshape(b, c, h * w).permute(0, 2, 1)
    flat = cross_attention(sd, p, flat, text_emb)
    return flat.permute(0, 2, 1).reshape(b, c, h, w)


def unet_forward(
    sd: SD,
    cfg,
    x: torch.Tensor,
    time: torch.Tensor,
    x_self_cond: Optional[torch.Tensor] = None,
    text_emb: Optional[torch.Tensor] = None,
    cond: Optional[torch.Tensor] = None,
    prefix: str = "",
    dropout_masks=None,
) -> torch.Tensor:
    """``dropout_masks`` (training mode): an iterable of dropout factors, one (B, C, H, W) tensor per Block in forward
    order, injected instead of nn.Dropout's own draws.

    DD/denoising_diffusion.py:349-390; text hooks
    DD/denoising_diffusion_text_conditional.py:131-214; the image-conditional
    variant concatenates ``cond`` in front of init_conv
    (DD/denoising_diffusion_image_conditional.py:51-55).

    ``cfg`` is a ``UnetConfig``-like object (dim, dim_mults, attn_heads, ...).
    """
    p = prefix
    drops = iter(dropout_masks) if dropout_masks is not None else None
    f = cfg.downsample_factor
    assert all(d % f == 0 for d in x.shape[-2:]), (
        f"your input dimensions {tuple(x.shape[-2:])} need to be divisible by {f}, given the unet"
    )
    hs = cfg.attn_heads  # one value or one per stage (cast_tuple, DD/denoising_diffusion.py:294; mid_attn: the last, :324)
    heads = tuple(hs) if isinstance(hs, (tuple, list)) else (hs,) * cfg.num_stages
    dh = cfg.attn_dim_head
    if cfg.self_condition:
        if x_self_cond is None:
            x_self_cond = torch.zeros_like(x)
        x = torch.cat((x_self_cond, x), dim=1)
    if cond is not None:
        x = torch.cat((x, cond), dim=1)

    x = F.conv2d(x, sd[p + "init_conv.weight"], sd[p + "init_conv.bias"], padding=3)
    r = x
    t = time_mlp(sd, p, time, cfg.dim, cfg.sinusoidal_pos_emb_theta)

    use_text = cfg.text_condition and text_emb is not None
    if use_text and not cfg.use_cross_attn:
        te = text_emb
        if te.dim() == 3 and te.size(1) == 1:
            te = te.squeeze(1)
        te = te.to(t.dtype)
        tf = F.linear(te, sd[p + "text_proj.0.weight"], sd[p + "text_proj.0.bias"])
        tf = F.linear(F.gelu(tf), sd[p + "text_proj.2.weight"], sd[p + "text_proj.2.bias"])
        t = F.linear(torch.cat((t, tf), dim=1), sd[p + "text_concat_proj.weight"], sd[p + "text_concat_proj.bias"])

    skips = []
    n = cfg.num_stages
    full = cfg.full_attn_
    for i in range(n):
        last = i >= n - 1
        q = f"{p}downs.{i}"
        x = resnet_block(sd, q + ".0", x, t, drops)
        skips.append(x)
        x = resnet_block(sd, q + ".1", x, t, drops)
        a = full_attention if full[i] else linear_attention
        x = a(sd, q + ".2", x, heads[i], dh) + x
        skips.append(x)
        if not last:
            x = downsample(sd, q + ".3", x)
        else:
            x = F.conv2d(x, sd[q + ".3.weight"], sd[q + ".3.bias"], padding=1)

    if use_text and cfg.use_cross_attn:
        x = _apply_cross(sd, p + "cross_attn_down", x, text_emb)
    x = resnet_block(sd, p + "mid_block1", x, t, drops)
    if use_text and cfg.use_cross_attn:
        x = _apply_cross(sd, p + "cross_attn", x, text_emb)
    x = full_attention(sd, p + "mid_attn", x, heads[-1], dh) + x
    x = resnet_block(sd, p + "mid_block2", x, t, drops)
    if use_text and cfg.use_cross_attn:
        x = _apply_cross(sd, p + "cross_attn_up", x, text_emb)

    for j in range(n):
        last = j == n - 1
        q = f"{p}ups.{j}"
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(sd, q + ".0", x, t, drops)
        x = torch.cat((x, skips.pop()), dim=1)
        x = resnet_block(sd, q + ".1", x, t, drops)
        a = full_attention if full[n - 1 - j] else linear_attention
        x = a(sd, q + ".2", x, heads[n - 1 - j], dh) + x
        if not last:
            x = upsample(sd, q + ".3", x)
        else:
            x = F.conv2d(x, sd[q + ".3.weight"], sd[q + ".3.bias"], padding=1)

    x = torch.cat((x, r), dim=1)
    x = resnet_block(sd, p + "final_res_block", x, t, drops)
    return F.conv2d(x, sd[p + "final_conv.weight"], sd[p + "final_conv.bias"])
