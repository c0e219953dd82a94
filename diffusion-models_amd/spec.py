"""Static description of the denoiser and of the noise schedule (host side).

Nothing in this module touches the GPU.  It answers three questions the rest of
the package (and the tests) need answered identically:

* which tensors a ``DenoisingDiffusion.state_dict()`` holds, with which names and
  shapes, for a given U-Net configuration            -> :func:`unet_param_spec`
* which 13 schedule buffers the diffusion wrapper registers -> :func:`make_schedule`
* which (t, t_next) pairs a DDIM run visits          -> :func:`ddim_time_pairs`

Reference behaviour restated here (paths relative to the reference checkout):
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion.py
    :233-343  Unet.__init__          (module tree -> parameter names/shapes)
    :399-433  linear / cosine / sigmoid beta schedules (fp64)
    :482-541  DenoisingDiffusion.__init__ buffers (fp64 -> fp32 once)
    :672-674  ddim time pairs
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion_text_conditional.py
    :38-52,:97-125  CrossAttention / text U-Net extra parameters
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

NUM_MEM_KV = 4  # LinearAttention / Attention default (denoising_diffusion.py:156,201)


@dataclass(frozen=True)
class UnetConfig:
    """Constructor arguments of the reference ``Unet`` that change shapes.

    Mirrors denoising_diffusion.py:234-252 (and the text subclass,
    denoising_diffusion_text_conditional.py:97).  Options the sampling path
    never uses (dropout at eval time, flash attention) are deliberately absent.  The random / learned sinusoidal
    embedding (:86-100) is here for ``Unet.forward`` alone: ``DenoisingDiffusion`` refuses such a U-Net (:456-457).
    """

    dim: int = 64
    init_dim: Optional[int] = None
    out_dim: Optional[int] = None
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    channels: int = 3
    self_condition: bool = False
    learned_variance: bool = False
    sinusoidal_pos_emb_theta: float = 10000.0
    learned_sinusoidal_cond: bool = False
    random_fourier_features: bool = False
    learned_sinusoidal_dim: int = 16
    attn_dim_head: int = 32
    attn_heads: object = 4  # int, or one value per stage (cast_tuple(attn_heads, num_stages), denoising_diffusion.py:294)
    full_attn: Optional[Tuple[bool, ...]] = None
    # image-conditional variant widens init_conv (denoising_diffusion_image_conditional.py:42-49)
    cond_channels: int = 0
    # text variant
    text_condition: bool = False
    use_cross_attn: bool = False
    text_emb_dim: int = 512

    # ---- derived quantities -------------------------------------------------
    @property
    def init_dim_(self) -> int:
        return self.init_dim if self.init_dim is not None else self.dim

    @property
    def input_channels(self) -> int:
        return self.channels * (2 if self.self_condition else 1) + self.cond_channels

    @property
    def out_dim_(self) -> int:
        if self.out_dim is not None:
            return self.out_dim
        return self.channels * (2 if self.learned_variance else 1)

    @property
    def time_dim(self) -> int:
        return self.dim * 4

    @property
    def random_or_learned_sinusoidal_cond(self) -> bool:
        return self.learned_sinusoidal_cond or self.random_fourier_features

    @property
    def fourier_dim(self) -> int:
        """Width of the time embedding in front of time_mlp.1 (:271-278)."""
        return self.learned_sinusoidal_dim + 1 if self.random_or_learned_sinusoidal_cond else self.dim

    @property
    def dims(self) -> List[int]:
        return [self.init_dim_, *[self.dim * m for m in self.dim_mults]]

    @property
    def in_out(self) -> List[Tuple[int, int]]:
        d = self.dims
        return list(zip(d[:-1], d[1:]))

    @property
    def num_stages(self) -> int:
        return len(self.dim_mults)

    @property
    def full_attn_(self) -> Tuple[bool, ...]:
        # denoising_diffusion.py:289-293: full attention only in the innermost stage
        if not self.full_attn:
            return (*((False,) * (self.num_stages - 1)), True)
        return tuple(self.full_attn)

    @property
    def attn_heads_(self) -> Tuple[int, ...]:
        """Heads of each stage's attention (:294, :310, :327); ``mid_attn`` takes the last entry (:324)."""
        h = self.attn_heads
        out = tuple(int(v) for v in h) if isinstance(h, (tuple, list)) else (int(h),) * self.num_stages
        assert len(out) == self.num_stages, "attn_heads needs one entry per stage"
        return out

    @property
    def hidden_dim(self) -> int:
        return self.attn_dim_head * self.attn_heads_[0]

    @property
    def downsample_factor(self) -> int:
        return 2 ** (self.num_stages - 1)


ParamSpec = List[Tuple[str, Tuple[int, ...]]]


def _resnet_spec(prefix: str, din: int, dout: int, time_dim: int) -> ParamSpec:
    s: ParamSpec = [
        (f"{prefix}.mlp.1.weight", (dout * 2, time_dim)),
        (f"{prefix}.mlp.1.bias", (dout * 2,)),
        (f"{prefix}.block1.proj.weight", (dout, din, 3, 3)),
        (f"{prefix}.block1.proj.bias", (dout,)),
        (f"{prefix}.block1.norm.g", (1, dout, 1, 1)),
        (f"{prefix}.block2.proj.weight", (dout, dout, 3, 3)),
        (f"{prefix}.block2.proj.bias", (dout,)),
        (f"{prefix}.block2.norm.g", (1, dout, 1, 1)),
    ]
    if din != dout:
        s += [
            (f"{prefix}.res_conv.weight", (dout, din, 1, 1)),
            (f"{prefix}.res_conv.bias", (dout,)),
        ]
    return s


def _attn_spec(prefix: str, dim: int, full: bool, heads: int, dim_head: int) -> ParamSpec:
    hidden = heads * dim_head
    if full:
        return [
            (f"{prefix}.mem_kv", (2, heads, NUM_MEM_KV, dim_head)),
            (f"{prefix}.norm.g", (1, dim, 1, 1)),
            (f"{prefix}.to_qkv.weight", (hidden * 3, dim, 1, 1)),
            (f"{prefix}.to_out.weight", (dim, hidden, 1, 1)),
            (f"{prefix}.to_out.bias", (dim,)),
        ]
    return [
        (f"{prefix}.mem_kv", (2, heads, dim_head, NUM_MEM_KV)),
        (f"{prefix}.norm.g", (1, dim, 1, 1)),
        (f"{prefix}.to_qkv.weight", (hidden * 3, dim, 1, 1)),
        (f"{prefix}.to_out.0.weight", (dim, hidden, 1, 1)),
        (f"{prefix}.to_out.0.bias", (dim,)),
        (f"{prefix}.to_out.1.g", (1, dim, 1, 1)),
    ]


def _cross_attn_spec(prefix: str, dim: int, ctx: int, heads: int, dim_head: int) -> ParamSpec:
    inner = heads * dim_head
    return [
        (f"{prefix}.to_q.weight", (inner, dim)),
        (f"{prefix}.to_k.weight", (inner, ctx)),
        (f"{prefix}.to_v.weight", (inner, ctx)),
        (f"{prefix}.to_out.0.weight", (dim, inner)),
        (f"{prefix}.to_out.0.bias", (dim,)),
        (f"{prefix}.to_out.1.g", (1, dim)),
    ]


def unet_param_spec(cfg: UnetConfig, prefix: str = "") -> ParamSpec:
    """(name, shape) of every U-Net parameter, in ``state_dict()`` order."""
    p = prefix
    td = cfg.time_dim
    spec: ParamSpec = [
        (f"{p}init_conv.weight", (cfg.init_dim_, cfg.input_channels, 7, 7)),
        (f"{p}init_conv.bias", (cfg.init_dim_,)),
        *([(f"{p}time_mlp.0.weights", (cfg.learned_sinusoidal_dim // 2,))] if cfg.random_or_learned_sinusoidal_cond else []),
        (f"{p}time_mlp.1.weight", (td, cfg.fourier_dim)),
        (f"{p}time_mlp.1.bias", (td,)),
        (f"{p}time_mlp.3.weight", (td, td)),
        (f"{p}time_mlp.3.bias", (td,)),
    ]
    n = len(cfg.in_out)
    for i, ((din, dout), full) in enumerate(zip(cfg.in_out, cfg.full_attn_)):
        last = i >= n - 1
        spec += _resnet_spec(f"{p}downs.{i}.0", din, din, td)
        spec += _resnet_spec(f"{p}downs.{i}.1", din, din, td)
        spec += _attn_spec(f"{p}downs.{i}.2", din, full, cfg.attn_heads_[i], cfg.attn_dim_head)
        if not last:
            spec += [(f"{p}downs.{i}.3.1.weight", (dout, din * 4, 1, 1)), (f"{p}downs.{i}.3.1.bias", (dout,))]
        else:
            spec += [(f"{p}downs.{i}.3.weight", (dout, din, 3, 3)), (f"{p}downs.{i}.3.bias", (dout,))]
    mid = cfg.dims[-1]
    for j, ((din, dout), full) in enumerate(zip(reversed(cfg.in_out), reversed(cfg.full_attn_))):
        last = j == n - 1
        spec += _resnet_spec(f"{p}ups.{j}.0", dout + din, dout, td)
        spec += _resnet_spec(f"{p}ups.{j}.1", dout + din, dout, td)
        spec += _attn_spec(f"{p}ups.{j}.2", dout, full, cfg.attn_heads_[n - 1 - j], cfg.attn_dim_head)
        if not last:
            spec += [(f"{p}ups.{j}.3.1.weight", (din, dout, 3, 3)), (f"{p}ups.{j}.3.1.bias", (din,))]
        else:
            spec += [(f"{p}ups.{j}.3.weight", (din, dout, 3, 3)), (f"{p}ups.{j}.3.bias", (din,))]
    # ModuleLists `downs` and `ups` are registered before the mid blocks (:306-307)
    spec += _resnet_spec(f"{p}mid_block1", mid, mid, td)
    spec += _attn_spec(f"{p}mid_attn", mid, True, cfg.attn_heads_[-1], cfg.attn_dim_head)
    spec += _resnet_spec(f"{p}mid_block2", mid, mid, td)
    spec += _resnet_spec(f"{p}final_res_block", cfg.init_dim_ * 2, cfg.init_dim_, td)
    spec += [
        (f"{p}final_conv.weight", (cfg.out_dim_, cfg.init_dim_, 1, 1)),
        (f"{p}final_conv.bias", (cfg.out_dim_,)),
    ]
    if cfg.text_condition and not cfg.use_cross_attn:
        spec += [
            (f"{p}text_proj.0.weight", (td, cfg.text_emb_dim)),
            (f"{p}text_proj.0.bias", (td,)),
            (f"{p}text_proj.2.weight", (td, td)),
            (f"{p}text_proj.2.bias", (td,)),
            (f"{p}text_concat_proj.weight", (td, td * 2)),
            (f"{p}text_concat_proj.bias", (td,)),
        ]
    if cfg.text_condition and cfg.use_cross_attn:
        for name in ("cross_attn", "cross_attn_down", "cross_attn_up"):
            spec += _cross_attn_spec(f"{p}{name}", mid, cfg.text_emb_dim, 4, cfg.attn_dim_head)
    return spec


# ----------------------------------------------------------------------------
# schedule
# ----------------------------------------------------------------------------

SCHEDULE_BUFFERS = (
    "betas",
    "alphas_cumprod",
    "alphas_cumprod_prev",
    "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod",
    "posterior_variance",
    "posterior_log_variance_clipped",
    "posterior_mean_coef1",
    "posterior_mean_coef2",
    "loss_weight",
)


def _betas(name: str, timesteps: int, **kw) -> torch.Tensor:
    f64 = torch.float64
    if name == "linear":  # denoising_diffusion.py:399-406
        scale = 1000 / timesteps
        return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=f64)
    steps = timesteps + 1
    t = torch.linspace(0, timesteps, steps, dtype=f64) / timesteps
    if name == "cosine":  # :408-418
        s = kw.get("s", 0.008)
        ac = torch.cos((t + s) / (1 + s) * math.pi * 0.5) ** 2
    elif name == "sigmoid":  # :420-433
        start, end, tau = kw.get("start", -3), kw.get("end", 3), kw.get("tau", 1)
        v_start = torch.tensor(start / tau).sigmoid()
        v_end = torch.tensor(end / tau).sigmoid()
        ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    else:
        raise ValueError(f"unknown beta schedule {name}")
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def make_schedule(timesteps: int = 1000, beta_schedule: str = "linear", *, ddpm: bool = True, objective: str = "pred_noise",
                  min_snr_loss_weight: bool = False, min_snr_gamma=5, **schedule_fn_kwargs) -> Dict[str, torch.Tensor]:
    """The 13 fp32 buffers of ``DenoisingDiffusion`` (denoising_diffusion.py:482-549).  ``loss_weight``: ones with
    ``ddpm=True`` (:532-533), else derived from the float64 signal-to-noise ratio and rounded to fp32 ONCE, as the
    reference's ``register_buffer`` does (:535-549)."""
    betas = _betas(beta_schedule, timesteps, **schedule_fn_kwargs)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    f64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(post_var.clamp(min=1e-20)),
        "posterior_mean_coef1": betas * torch.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac),
        "loss_weight": torch.ones(timesteps, dtype=torch.float64),
    }
    if not ddpm:
        snr = ac / (1 - ac)
        clipped = snr.clone()
        if min_snr_loss_weight:
            clipped.clamp_(max=min_snr_gamma)
        if objective not in ("pred_noise", "pred_x0", "pred_v"):
            raise ValueError(f"unknown objective {objective}")
        f64["loss_weight"] = {"pred_noise": clipped / snr, "pred_x0": clipped, "pred_v": clipped / (snr + 1)}[objective]
    return {k: v.to(torch.float32) for k, v in f64.items()}


def ddim_time_pairs(total_timesteps: int, sampling_timesteps: int) -> List[Tuple[int, int]]:
    """[(t, t_next)] as visited by ddim_sample (denoising_diffusion.py:672-674)."""
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


DM_COEFS = 8  # floats per step row handed to dm_sample (include/dm_hip.h)


def ddpm_step_table(sched: Dict[str, torch.Tensor]) -> Tuple[List[int], torch.Tensor]:
    """Per-step scalars of p_sample_loop, computed with the reference's own fp32
    tensor arithmetic (denoising_diffusion.py:570-574, :594-601, :643-644).
    Row i (t = T-1-i): [sqrt_recip_ac, sqrt_recipm1_ac, coef1, coef2, exp(0.5*logvar), t>0, sqrt_ac, sqrt_1m_ac]
    (the last two feed predict_start_from_v, :588-592, objective pred_v)."""
    T = int(sched["betas"].shape[0])
    times = list(reversed(range(T)))
    idx = torch.tensor(times)
    c = torch.zeros(T, DM_COEFS, dtype=torch.float32)
    c[:, 0] = sched["sqrt_recip_alphas_cumprod"][idx]
    c[:, 1] = sched["sqrt_recipm1_alphas_cumprod"][idx]
    c[:, 2] = sched["posterior_mean_coef1"][idx]
    c[:, 3] = sched["posterior_mean_coef2"][idx]
    c[:, 4] = (0.5 * sched["posterior_log_variance_clipped"][idx]).exp()
    c[:, 5] = (idx > 0).to(torch.float32)
    c[:, 6] = sched["sqrt_alphas_cumprod"][idx]
    c[:, 7] = sched["sqrt_one_minus_alphas_cumprod"][idx]
    return times, c


def ddim_step_table(sched: Dict[str, torch.Tensor], sampling_timesteps: int, eta: float) -> Tuple[List[int], torch.Tensor]:
    """Per-step scalars of ddim_sample (denoising_diffusion.py:684-701) on 0-dim fp32 tensors.
    Row i: [sqrt_recip_ac[t], sqrt_recipm1_ac[t], sqrt(alpha_next), c, sigma, t_next>=0, sqrt_ac[t], sqrt_1m_ac[t]]."""
    T = int(sched["betas"].shape[0])
    pairs = ddim_time_pairs(T, sampling_timesteps)
    ac = sched["alphas_cumprod"]
    c = torch.zeros(len(pairs), DM_COEFS, dtype=torch.float32)
    for i, (t, tn) in enumerate(pairs):
        c[i, 0] = sched["sqrt_recip_alphas_cumprod"][t]
        c[i, 1] = sched["sqrt_recipm1_alphas_cumprod"][t]
        c[i, 6] = sched["sqrt_alphas_cumprod"][t]
        c[i, 7] = sched["sqrt_one_minus_alphas_cumprod"][t]
        if tn < 0:
            continue  # flag 0: img = x_start (:686-689)
        alpha, alpha_next = ac[t], ac[tn]
        sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        cc = (1 - alpha_next - sigma ** 2).sqrt()
        c[i, 2] = alpha_next.sqrt()
        c[i, 3] = cc
        c[i, 4] = sigma
        c[i, 5] = 1.0
    return [p[0] for p in pairs], c


# ----------------------------------------------------------------------------
# VAE decoder (LDM)
# ----------------------------------------------------------------------------


@dataclass(frozen=True)
class DecoderConfig:
    """``ddconfig`` keys the Decoder reads (latent-diffusion/ldm/modules/diffusionmodules/model.py:476-550)."""

    ch: int = 64
    out_ch: int = 3
    ch_mult: Tuple[int, ...] = (1, 2)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = ()
    resolution: int = 32
    z_channels: int = 3
    embed_dim: int = 3  # VQModel.post_quant_conv: embed_dim -> z_channels (autoencoder.py:53)

    @property
    def num_resolutions(self) -> int:
        return len(self.ch_mult)

    @property
    def z_res(self) -> int:
        return self.resolution // 2 ** (self.num_resolutions - 1)


def _vae_resblock_spec(prefix: str, cin: int, cout: int) -> ParamSpec:
    s: ParamSpec = [
        (f"{prefix}.norm1.weight", (cin,)),
        (f"{prefix}.norm1.bias", (cin,)),
        (f"{prefix}.conv1.weight", (cout, cin, 3, 3)),
        (f"{prefix}.conv1.bias", (cout,)),
        (f"{prefix}.norm2.weight", (cout,)),
        (f"{prefix}.norm2.bias", (cout,)),
        (f"{prefix}.conv2.weight", (cout, cout, 3, 3)),
        (f"{prefix}.conv2.bias", (cout,)),
    ]
    if cin != cout:
        s += [(f"{prefix}.nin_shortcut.weight", (cout, cin, 1, 1)), (f"{prefix}.nin_shortcut.bias", (cout,))]
    return s


def _vae_attn_spec(prefix: str, c: int) -> ParamSpec:
    s: ParamSpec = [(f"{prefix}.norm.weight", (c,)), (f"{prefix}.norm.bias", (c,))]
    for n in ("q", "k", "v", "proj_out"):
        s += [(f"{prefix}.{n}.weight", (c, c, 1, 1)), (f"{prefix}.{n}.bias", (c,))]
    return s


def decoder_param_spec(cfg: DecoderConfig, prefix: str = "") -> ParamSpec:
    """Parameters of ``VQModel.post_quant_conv`` + ``VQModel.decoder``."""
    p = prefix
    block_in = cfg.ch * cfg.ch_mult[-1]
    spec: ParamSpec = [
        (f"{p}post_quant_conv.weight", (cfg.z_channels, cfg.embed_dim, 1, 1)),
        (f"{p}post_quant_conv.bias", (cfg.z_channels,)),
        (f"{p}decoder.conv_in.weight", (block_in, cfg.z_channels, 3, 3)),
        (f"{p}decoder.conv_in.bias", (block_in,)),
    ]
    spec += _vae_resblock_spec(f"{p}decoder.mid.block_1", block_in, block_in)
    spec += _vae_attn_spec(f"{p}decoder.mid.attn_1", block_in)
    spec += _vae_resblock_spec(f"{p}decoder.mid.block_2", block_in, block_in)
    curr_res = cfg.z_res
    per_level: Dict[int, ParamSpec] = {}
    for lvl in reversed(range(cfg.num_resolutions)):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        s: ParamSpec = []
        attn: ParamSpec = []
        for b in range(cfg.num_res_blocks + 1):
            s += _vae_resblock_spec(f"{p}decoder.up.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if curr_res in cfg.attn_resolutions:
                attn += _vae_attn_spec(f"{p}decoder.up.{lvl}.attn.{b}", block_in)
        s += attn
        if lvl != 0:
            s += [
                (f"{p}decoder.up.{lvl}.upsample.conv.weight", (block_in, block_in, 3, 3)),
                (f"{p}decoder.up.{lvl}.upsample.conv.bias", (block_in,)),
            ]
            curr_res *= 2
        per_level[lvl] = s
    for lvl in range(cfg.num_resolutions):  # ModuleList order after the insert(0, …) calls
        spec += per_level[lvl]
    spec += [
        (f"{p}decoder.norm_out.weight", (block_in,)),
        (f"{p}decoder.norm_out.bias", (block_in,)),
        (f"{p}decoder.conv_out.weight", (cfg.out_ch, block_in, 3, 3)),
        (f"{p}decoder.conv_out.bias", (cfg.out_ch,)),
    ]
    return spec


@dataclass(frozen=True)
class EncoderConfig:
    """``ddconfig`` keys the Encoder reads (latent-diffusion/ldm/modules/diffusionmodules/model.py:385-449) plus the
    VQModel's codebook shape (latent-diffusion/ldm/models/autoencoder.py:31-49)."""

    ch: int = 64
    in_channels: int = 3
    ch_mult: Tuple[int, ...] = (1, 2)
    num_res_blocks: int = 2
    attn_resolutions: Tuple[int, ...] = ()
    resolution: int = 32
    z_channels: int = 3
    embed_dim: int = 3
    n_embed: int = 8192
    double_z: bool = False

    @property
    def num_resolutions(self) -> int:
        return len(self.ch_mult)


def encoder_param_spec(cfg: EncoderConfig, prefix: str = "") -> ParamSpec:
    """Parameters of ``VQModel.encoder`` + ``quant_conv`` + ``quantize.embedding`` in ``state_dict()`` order of the
    Encoder (down.{l}.block.*, down.{l}.attn.*, down.{l}.downsample, mid, norm_out, conv_out)."""
    p = prefix
    spec: ParamSpec = [
        (f"{p}encoder.conv_in.weight", (cfg.ch, cfg.in_channels, 3, 3)),
        (f"{p}encoder.conv_in.bias", (cfg.ch,)),
    ]
    curr_res = cfg.resolution
    in_ch_mult = (1,) + tuple(cfg.ch_mult)
    block_in = cfg.ch
    for lvl in range(cfg.num_resolutions):
        block_in = cfg.ch * in_ch_mult[lvl]
        block_out = cfg.ch * cfg.ch_mult[lvl]
        attn: ParamSpec = []
        for b in range(cfg.num_res_blocks):
            spec += _vae_resblock_spec(f"{p}encoder.down.{lvl}.block.{b}", block_in, block_out)
            block_in = block_out
            if curr_res in cfg.attn_resolutions:
                attn += _vae_attn_spec(f"{p}encoder.down.{lvl}.attn.{b}", block_in)
        spec += attn
        if lvl != cfg.num_resolutions - 1:
            spec += [
                (f"{p}encoder.down.{lvl}.downsample.conv.weight", (block_in, block_in, 3, 3)),
                (f"{p}encoder.down.{lvl}.downsample.conv.bias", (block_in,)),
            ]
            curr_res //= 2
    spec += _vae_resblock_spec(f"{p}encoder.mid.block_1", block_in, block_in)
    spec += _vae_attn_spec(f"{p}encoder.mid.attn_1", block_in)
    spec += _vae_resblock_spec(f"{p}encoder.mid.block_2", block_in, block_in)
    zc = 2 * cfg.z_channels if cfg.double_z else cfg.z_channels
    spec += [
        (f"{p}encoder.norm_out.weight", (block_in,)),
        (f"{p}encoder.norm_out.bias", (block_in,)),
        (f"{p}encoder.conv_out.weight", (zc, block_in, 3, 3)),
        (f"{p}encoder.conv_out.bias", (zc,)),
        (f"{p}quant_conv.weight", (cfg.embed_dim, zc, 1, 1)),
        (f"{p}quant_conv.bias", (cfg.embed_dim,)),
        (f"{p}quantize.embedding.weight", (cfg.n_embed, cfg.embed_dim)),
    ]
    return spec

