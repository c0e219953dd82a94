"""Host-side mirror of the reference diffusion wrappers: same method names,
arguments, return shapes and assertions, with the loop executed by
``dm_sample`` in libdm_hip.so (hipGraph-replayed denoise step).

Reference surface mirrored (paths relative to the reference checkout):
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion.py
      :435-568  DenoisingDiffusion.__init__ / buffers / device
      :638-664  p_sample, p_sample_loop      :666-708 ddim_sample     :779-783 sample
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion_text_conditional.py
      :264-453  TextConditionalDenoisingDiffusion (text_emb threaded through the loop)
  latent-diffusion/ldm/models/latent_diffusion.py:9-66  LatentDiffusion

Extensions (keyword-only, default to the reference behaviour): ``noise`` injects
the N(0,1) draws (a callable ``shape -> cpu tensor`` called in the reference's
draw order, e.g. ``oracle.sampler_oracle.NoiseStream``) for parity tests;
``seed`` selects the device Philox stream used otherwise (default: drawn from
torch's global CPU generator, so ``torch.manual_seed`` makes ``sample()``
reproducible as it does for the reference -- the noise VALUES differ from
``torch.randn``'s, the control does not); ``sample_offset`` is the index of the
call's first sample in a global batch sharded over ranks (``dist.sample_sharded``):
Philox counters are global element indices, so the concatenated shards equal
the unsharded batch bit for bit.
"""
from __future__ import annotations

from collections import namedtuple

import ctypes as C
import os
import pickle
import random
from pathlib import Path
from typing import Callable, List, Optional, Sequence, Tuple

import torch

from . import _lib
from .spec import SCHEDULE_BUFFERS, ddim_step_table, ddpm_step_table, make_schedule

DDPM, DDIM = 0, 1


def _identity(t, *a, **k):
    return t


def _default_seed() -> int:
    """A Philox key from torch's global CPU generator (reproducible under torch.manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


ModelPrediction = namedtuple("ModelPrediction", ["pred_noise", "pred_x_start"])  # :33


class DenoisingDiffusion:
    def __init__(
        self,
        model,
        *,
        image_size,
        timesteps=1000,
        sampling_timesteps=None,
        objective="pred_noise",
        beta_schedule="linear",
        schedule_fn_kwargs=dict(),
        ddim_sampling_eta=0.0,
        auto_normalize=True,
        use_graph=True,
        offset_noise_strength=0.0,
        min_snr_loss_weight=False,
        min_snr_gamma=5,
        immiscible=False,
        ddpm=True,
        hybrid_loss=False,
    ):
        assert not (type(self) == DenoisingDiffusion and model.channels != model.out_dim)
        assert not getattr(model, "random_or_learned_sinusoidal_cond", False)
        self.model = model
        self.channels = model.channels
        self.self_condition = model.self_condition
        if isinstance(image_size, int):
            image_size = (image_size, image_size)
        assert isinstance(image_size, (tuple, list)) and len(image_size) == 2, (
            "image size must be a integer or a tuple/list of two integers"
        )
        self.image_size = tuple(image_size)
        assert objective in {"pred_noise", "pred_x0", "pred_v"}, "objective must be pred_noise, pred_x0 or pred_v"
        self.objective = objective
        self._objective_id = _lib.OBJECTIVES[objective]
        # raises ValueError on an unknown schedule name; loss_weight (ones, or from the float64 SNR, :532-549) comes with it
        sched = make_schedule(timesteps, beta_schedule, ddpm=ddpm, objective=objective, min_snr_loss_weight=min_snr_loss_weight,
                              min_snr_gamma=min_snr_gamma, **schedule_fn_kwargs)
        self.num_timesteps = int(sched["betas"].shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else self.num_timesteps
        assert self.sampling_timesteps <= self.num_timesteps
        self.is_ddim_sampling = self.sampling_timesteps < self.num_timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self.offset_noise_strength = offset_noise_strength
        self.immiscible, self.hybrid_loss = immiscible, hybrid_loss
        self._sched = sched  # fp32 CPU tensors; the per-step scalars are derived from them on the host
        for k, v in sched.items():
            setattr(self, k, v)
        self.normalize = (lambda img: img * 2 - 1) if auto_normalize else _identity
        self.unnormalize = (lambda t: (t + 1) * 0.5) if auto_normalize else _identity
        self._unnormalize_flag = 1 if auto_normalize else 0
        self.use_graph = use_graph
        self._lib = _lib.load()

    # -- module-ish surface the reference's callers touch -------------------------------------------
    @property
    def device(self):
        return self.model.device

    def eval(self):
        return self

    def to(self, device=None, *args, **kwargs):
        """``module.to(device)`` of the reference's scripts: the handle was created on its device (``device=`` of the
        constructor) and cannot move; the same device (or a dtype / no device) is accepted and ignored."""
        if isinstance(device, (str, torch.device)) and torch.device(device).type == "cuda":
            want = torch.device(device)
            have = torch.device(self.device)
            if want.index is not None and want.index != (have.index or 0):
                raise RuntimeError(f"this object lives on {have}; construct it with device={want!s}")
        elif isinstance(device, (str, torch.device)) and torch.device(device).type != "cuda":
            raise RuntimeError("the HIP path has no CPU fallback")
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{device}" if isinstance(device, int) else device)

    def parameters(self):
        """The U-Net's parameters (and the frozen VAE's, for the LDM classes, when it exposes them), as the scripts count them."""
        yield from self.model.parameters()
        vae = getattr(self, "vae", None)
        if vae is not None and hasattr(vae, "parameters"):
            yield from vae.parameters()

    def sample_shape(self):
        """(C, H, W) of one sample as ``sample()`` returns it (``dist.sample_global`` builds empty shards from it)."""
        (h, w), c = self.image_size, self.channels
        vae = getattr(self, "vae", None)
        return tuple(vae.decoded_shape((c, h, w))) if vae is not None else (c, h, w)

    def state_dict(self):
        """``DenoisingDiffusion.state_dict()`` of the reference module: the 13 schedule buffers, then ``model.*`` (the U-Net's
        current parameters: the device-resident training state in training mode, else the values that were loaded)."""
        out = {k: getattr(self, k) for k in SCHEDULE_BUFFERS}
        if getattr(self.model, "_loaded", False):
            out.update({"model." + k: v for k, v in self.model.state_dict().items()})
        return out

    def load_state_dict(self, state_dict, strict=True):
        """Accepts ``DenoisingDiffusion.state_dict()``: 13 schedule buffers + ``model.*``."""
        model_sd = {k[len("model."):]: v for k, v in state_dict.items() if k.startswith("model.")}
        for k in SCHEDULE_BUFFERS:
            if k in state_dict:
                if tuple(state_dict[k].shape) != tuple(self._sched[k].shape):
                    raise RuntimeError(f"size mismatch for {k}")
                self._sched[k] = state_dict[k].detach().to("cpu", torch.float32).clone()
                setattr(self, k, self._sched[k])
            elif strict:
                raise RuntimeError(f"missing schedule buffer {k}")
        self.model.load_state_dict(model_sd, strict=strict)
        return self

    # -- per-step scalars, computed as the reference computes them (fp32 tensor arithmetic) --------
    def _ddpm_tables(self) -> Tuple[List[int], torch.Tensor]:
        return ddpm_step_table(self._sched)

    def _ddim_tables(self, S: int) -> Tuple[List[int], torch.Tensor]:
        return ddim_step_table(self._sched, S, self.ddim_sampling_eta)

    # -- the loop --------------------------------------------------------------------------------------
    def _randn(self, shape, seed: int, draw: int, sample_offset: int = 0) -> torch.Tensor:
        out = torch.empty(tuple(shape), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        per_sample = out.numel() // max(int(shape[0]), 1)
        _lib.check(self._lib.dm_randn(_lib.ptr(out), out.numel(), C.c_uint64(seed), C.c_uint64(draw),
                                      C.c_uint64(int(sample_offset) * per_sample), stream))
        return out

    def _run(self, kind, shape, times, coefs, takes_noise: Sequence[bool], return_all_timesteps, noise, seed,
             text_emb=None, max_steps=None, cond=None, sample_offset=0, x_init=None, unnormalize=None):
        shape = tuple(int(v) for v in shape)
        B, Cc, H, W = shape
        assert Cc == self.channels, f"shape has {Cc} channels, the model {self.channels}"
        f = self.model.downsample_factor
        assert B > 0 and H % f == 0 and W % f == 0, f"shape {shape}: the sides must be divisible by {f}"
        n_steps = len(times)
        if seed is None:
            seed = _default_seed()
        sample_offset = int(sample_offset)
        if noise is not None:
            x_T = (noise(shape) if x_init is None else x_init).to(self.device, torch.float32).contiguous()
            rows = []
            zero = None
            for flag in takes_noise:
                if flag:
                    rows.append(noise(shape).to(torch.float32))
                else:
                    zero = zero if zero is not None else torch.zeros(shape, dtype=torch.float32)
                    rows.append(zero)
            noise_dev = torch.stack(rows, dim=0).to(self.device).contiguous()
        else:
            x_T = (self._randn(shape, seed, 0, sample_offset) if x_init is None
                   else x_init.to(self.device, torch.float32).contiguous())
            noise_dev = None
        assert tuple(x_T.shape) == shape, f"initial state {tuple(x_T.shape)} does not match {shape}"
        assert noise_dev is None or tuple(noise_dev.shape[1:]) == shape, "noise() must return tensors of the sampled shape"
        if max_steps is not None:  # bounded run (bench / smoke): first `max_steps` iterations only
            n_steps = min(n_steps, int(max_steps))
        ctx, m = (None, 0)
        if text_emb is not None:
            ctx, m = self.model._ctx(text_emb, B)
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        all_steps = (torch.empty((n_steps + 1,) + shape, device=self.device, dtype=torch.float32)
                     if return_all_timesteps else None)
        times_arr = (C.c_int64 * n_steps)(*[int(t) for t in times[:n_steps]])
        coefs = coefs[:n_steps].contiguous()
        coefs_ptr = C.cast(coefs.data_ptr(), C.POINTER(C.c_float))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        a = _lib.SampleArgs()
        a.kind, a.objective, a.self_condition, a.n_steps = kind, self._objective_id, int(bool(self.self_condition)), n_steps
        a.times_host, a.coefs_host = C.cast(times_arr, C.POINTER(C.c_int64)), coefs_ptr
        a.x_T, a.noise, a.seed, a.sample_offset = _lib.ptr(x_T), _lib.ptr(noise_dev), seed, sample_offset
        a.ctx, a.ctx_tokens = _lib.ptr(ctx), m
        if cond is not None:
            cond = cond.to(self.device, torch.float32).contiguous()
            assert cond.shape[0] == B and tuple(cond.shape[2:]) == (H, W), "batch / size mismatch between x and cond"
            a.cond, a.cond_channels = _lib.ptr(cond), int(cond.shape[1])
        a.out, a.all_steps = _lib.ptr(out), _lib.ptr(all_steps)
        a.B, a.H, a.W = B, H, W
        a.unnormalize = self._unnormalize_flag if unnormalize is None else int(bool(unnormalize))
        a.use_graph, a.stream = 1 if self.use_graph else 0, stream
        _lib.check(self._lib.dm_sample_ex(self.model._handle, C.byref(a)))
        if not return_all_timesteps:
            return out
        ret = all_steps.permute(1, 0, 2, 3, 4).contiguous()  # (B, n_steps+1, C, H, W) like torch.stack(imgs, dim=1)
        return self.unnormalize(ret)

    # -- training half (denoising_diffusion.py:805-900): loss and gradients in libdm_hip.so --------------------------
    def train(self, mode: bool = True):
        self.model.train(mode)
        return self

    def _tcoef(self, t: torch.Tensor) -> torch.Tensor:
        """(B, 12): what `extract` gathers for q_sample / predict_v / the loss weight / predict_start_from_noise and, for the
        hybrid KL term, q_posterior at each sample's timestep (DM_TRAIN_COEFS rows of include/dm_hip.h)."""
        t = t.detach().to("cpu", torch.long)
        s = self._sched
        z = torch.zeros(t.shape[0])
        return torch.stack([s["sqrt_alphas_cumprod"][t], s["sqrt_one_minus_alphas_cumprod"][t], s["loss_weight"][t],
                            (t > 0).float(), s["sqrt_recip_alphas_cumprod"][t], s["sqrt_recipm1_alphas_cumprod"][t], z, z,
                            s["posterior_mean_coef1"][t], s["posterior_mean_coef2"][t], s["posterior_variance"][t],
                            s["posterior_log_variance_clipped"][t]], dim=1).to(torch.float32).contiguous()

    def noise_assignment(self, x_start, noise):
        """:805-809 (immiscible diffusion): the assignment of noise rows to images that minimises the total L2 distance.
        The (B, B) distance matrix is formed on the device; ``scipy.optimize.linear_sum_assignment`` runs on the host, as
        in the reference."""
        from scipy.optimize import linear_sum_assignment

        b = x_start.shape[0]
        x_start = x_start.to(self.device, torch.float32).contiguous()
        noise = noise.to(self.device, torch.float32).contiguous()
        dist = torch.empty((b, b), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_op_cdist(_lib.ptr(x_start), _lib.ptr(noise), _lib.ptr(dist), b, b, x_start[0].numel(), stream))
        _, assign = linear_sum_assignment(dist.cpu().numpy())
        return torch.from_numpy(assign).long()

    def _assigned(self, x_start, noise):
        """noise[assign] of :815-817 as a new device tensor."""
        assign = self.noise_assignment(x_start, noise).contiguous()
        out = torch.empty_like(noise)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_op_gather_rows(_lib.ptr(noise), C.cast(assign.data_ptr(), C.POINTER(C.c_int64)), _lib.ptr(out),
                                               noise.shape[0], noise[0].numel(), stream))
        torch.cuda.current_stream(self.device).synchronize()  # `assign` is host memory read by the enqueueing call only
        return out

    def q_sample(self, x_start, t, noise=None):
        """:813-821, with the immiscible noise assignment of :815-817 when the object was built with ``immiscible=True``."""
        x_start = x_start.to(self.device, torch.float32).contiguous()
        noise = (noise.to(self.device, torch.float32).contiguous() if noise is not None
                 else self._randn(x_start.shape, _default_seed(), 0))
        if self.immiscible:
            noise = self._assigned(x_start, noise)
        coef = self._tcoef(t)
        out = torch.empty_like(x_start)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_op_q_sample(_lib.ptr(x_start), _lib.ptr(noise), C.cast(coef.data_ptr(), C.POINTER(C.c_float)),
                                            _lib.ptr(out), x_start.shape[0], x_start[0].numel(), stream))
        return out

    def p_losses(self, x_start, t, noise=None, offset_noise_strength=None, cond=None, *, return_model_out=False,
                 loss_scale=1.0, accumulate=False, self_cond=None, text_emb=None, offset_noise=None, sync=True):
        """:823-889: returns the loss (0-dim CPU tensor); the parameter gradients stay on the model
        (``self.model.grad(name)`` / ``.grads()``) -- loss and backward are one call of the library, there is no autograd
        graph to keep.  ``loss_scale`` / ``accumulate`` are the micro-batch loop of ``Trainer.train`` (:1164-1176):
        ``loss / gradient_accumulate_every`` with the gradients added up.  With ``Unet(self_condition=True)`` half of the
        calls condition on a gradient-free prediction of x_start (:846-855; ``self_cond=True / False`` forces the branch).
        Offset noise (:830-834; ``offset_noise``: the (B, C) draw, for tests) and the immiscible noise assignment
        (:815-817: q_sample mixes in ``noise[assign]`` while the target stays the unpermuted ``noise``, as in the
        reference) are on this path, and so is the hybrid (KL) term (:880-897, ``hybrid_loss=True``): the reference evaluates it
        through ``p_mean_variance``, a second forward pass with gradients -- without dropout that pass repeats the first bit
        for bit and both terms share one pass here; with dropout it draws new masks, so the KL term runs as a second,
        accumulating call.  As in the reference, a batch that holds a ``t = 0`` sample has a NaN loss (the KL expression
        divides by ``posterior_variance[0] = 0`` before the mask multiplies).  ``sync=False`` returns the loss as a 0-dim DEVICE tensor
        without waiting for the GPU -- what the reference's loss is until ``Trainer`` calls ``.item()`` on it (:1173)."""
        if offset_noise_strength is None:
            offset_noise_strength = self.offset_noise_strength
        sc_mode = 0
        if self.self_condition:
            # :846-855: half of the iterations condition on the x_start a gradient-free pass predicts (self_cond= forces it)
            use = (random.random() < 0.5) if self_cond is None else bool(self_cond)
            sc_mode = 2 if use else 1
        if not getattr(self.model, "_training", False):
            self.model.train()
        x_start = x_start.to(self.device, torch.float32).contiguous()
        b, c, h, w = x_start.shape
        f = self.model.downsample_factor
        if c != self.channels or h % f or w % f:
            raise RuntimeError(f"x_start {tuple(x_start.shape)}: expected {self.channels} channels and sides divisible by {f}")
        noise = (noise.to(self.device, torch.float32).contiguous() if noise is not None
                 else self._randn(x_start.shape, _default_seed(), 0))
        if noise.shape != x_start.shape or t.numel() != b:
            raise RuntimeError(f"noise {tuple(noise.shape)} / t ({t.numel()} entries) do not match x_start {tuple(x_start.shape)}")
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if offset_noise_strength and offset_noise_strength > 0.0:
            offs = (offset_noise.to(self.device, torch.float32).contiguous() if offset_noise is not None
                    else self._randn((b, c), _default_seed(), 3))
            assert tuple(offs.shape) == (b, c)
            noise = noise.clone()  # the caller's tensor stays as it was
            _lib.check(self._lib.dm_op_offset_noise(_lib.ptr(noise), _lib.ptr(offs), float(offset_noise_strength), b * c, h * w,
                                                    stream))
        noise_q = self._assigned(x_start, noise) if self.immiscible else None
        t_cpu = t.detach().to("cpu", torch.long).contiguous()
        coef = self._tcoef(t_cpu)
        cc = 0
        if cond is not None:  # image-conditional variant (denoising_diffusion_image_conditional.py:251-311)
            cond = cond.to(self.device, torch.float32).contiguous()
            assert cond.shape[0] == b and tuple(cond.shape[2:]) == (h, w), "batch / size mismatch between x and cond"
            cc = int(cond.shape[1])
        ctx, m = self.model._ctx(text_emb, b) if text_emb is not None else (None, 0)
        loss = C.c_float(0.0)
        out = torch.empty_like(x_start) if return_model_out else None
        t_arr = (C.c_int64 * b)(*[int(v) for v in t_cpu.tolist()])
        a = _lib.TrainArgs()
        a.x_start, a.noise, a.noise_q, a.cond, a.ctx = (_lib.ptr(v) for v in (x_start, noise, noise_q, cond, ctx))
        a.t_host = C.cast(t_arr, C.POINTER(C.c_int64))
        a.coef_host, a.coef_stride = C.cast(coef.data_ptr(), C.POINTER(C.c_float)), int(coef.shape[1])
        a.cond_channels, a.ctx_tokens, a.self_cond, a.objective = cc, m, sc_mode, self._objective_id
        a.loss_scale, a.accumulate = float(loss_scale), int(bool(accumulate))
        a.loss_out_host = C.pointer(loss) if sync else None
        a.model_out, a.B, a.H, a.W, a.stream = _lib.ptr(out), b, h, w, stream
        a.loss_terms = 1
        if self.hybrid_loss:
            if cond is not None:
                # the reference's image-conditional p_losses calls p_mean_variance WITHOUT cond (denoising_diffusion_image_
                # conditional.py:295): its 6-channel U-Net then receives 3 channels and raises
                raise NotImplementedError("hybrid_loss with an image condition fails in the reference (p_mean_variance is "
                                          "called without cond); not reproduced")
            mask_sum = (t_cpu > 0).float().sum()  # fp32, as the reference forms kl / (mask.sum() + 1e-8), :893-895
            a.kl_scale = float(torch.tensor(0.001, dtype=torch.float32) / (mask_sum + 1e-8))
            a.loss_terms = 3 if float(getattr(self.model, "dropout", 0.0) or 0.0) == 0.0 else 1
        _lib.check(self._lib.dm_unet_loss_backward_ex(self.model._handle, C.byref(a)))
        if self.hybrid_loss and a.loss_terms == 1:
            # dropout: p_mean_variance's forward pass draws its own masks -- a second, accumulating call for the KL term
            mse = C.c_float(loss.value)
            if not sync:
                mse_dev = torch.empty((), device=self.device, dtype=torch.float32)
                _lib.check(self._lib.dm_unet_train_scalar(self.model._handle, 0, _lib.ptr(mse_dev), stream))
            a.loss_terms, a.accumulate, a.model_out = 2, 1, None
            _lib.check(self._lib.dm_unet_loss_backward_ex(self.model._handle, C.byref(a)))
            if sync:
                loss = C.c_float(mse.value + loss.value)
        if sync:
            val = torch.tensor(loss.value, dtype=torch.float32)
        else:
            val = torch.empty((), device=self.device, dtype=torch.float32)
            _lib.check(self._lib.dm_unet_train_scalar(self.model._handle, 0, _lib.ptr(val), stream))
            if self.hybrid_loss and a.loss_terms == 2:
                val = val + mse_dev
        return (val, out) if return_model_out else val

    def forward(self, img, *args, **kwargs):
        """:892-899: random timesteps, normalise, p_losses."""
        b, c, h, w = img.shape
        assert (h, w) == tuple(self.image_size), f"height and width of image must be {self.image_size}"
        t = torch.randint(0, self.num_timesteps, (b,)).long()
        return self.p_losses(self.normalize(img.to(self.device, torch.float32)), t, *args, **kwargs)

    __call__ = forward

    @torch.inference_mode()
    def p_sample_loop(self, shape, return_all_timesteps=False, *, noise=None, seed=None, max_steps=None, text_emb=None,
                      sample_offset=0):
        times, coefs = self._ddpm_tables()
        takes = [t > 0 for t in times]
        return self._run(DDPM, shape, times, coefs, takes, return_all_timesteps, noise, seed, text_emb, max_steps,
                         sample_offset=sample_offset)

    @torch.inference_mode()
    def ddim_sample(self, shape, sampling_timesteps=None, return_all_timesteps=False, *, noise=None, seed=None,
                    max_steps=None, text_emb=None, sample_offset=0):
        if sampling_timesteps is None:
            sampling_timesteps = self.sampling_timesteps
        times, coefs = self._ddim_tables(sampling_timesteps)
        takes = [bool(c[5] != 0) for c in coefs]
        return self._run(DDIM, shape, times, coefs, takes, return_all_timesteps, noise, seed, text_emb, max_steps,
                         sample_offset=sample_offset)

    @torch.inference_mode()
    def sample(self, batch_size=16, return_all_timesteps=False, **kw):
        (h, w), channels = self.image_size, self.channels
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        return sample_fn((batch_size, channels, h, w), return_all_timesteps=return_all_timesteps, **kw)

    @torch.inference_mode()
    def interpolate(self, x1, x2, t=None, lam=0.5, *, noise=None, seed=None):
        """:786-803: diffuse x1 and x2 to step t, mix them with weight lam, run the reverse loop from t - 1 down to 0.
        Like the reference it returns the image as the loop leaves it (no unnormalize).  ``noise`` draws, in the
        reference's order: q_sample(x1), q_sample(x2), then one per reverse step with i > 0."""
        return self._interpolate(x1, x2, t, lam, noise, seed)

    def _interpolate(self, x1, x2, t, lam, noise, seed, text_emb=None, cond=None):
        assert x1.shape == x2.shape
        b = x1.shape[0]
        t = self.num_timesteps - 1 if t is None else int(t)
        tb = torch.full((b,), t, dtype=torch.long)
        if seed is None:
            seed = _default_seed()
        n1 = noise(tuple(x1.shape)) if noise is not None else self._randn(x1.shape, seed, 1 << 20)
        n2 = noise(tuple(x1.shape)) if noise is not None else self._randn(x1.shape, seed, (1 << 20) + 1)
        xt1, xt2 = self.q_sample(x1, tb, n1), self.q_sample(x2, tb, n2)
        img = (1 - lam) * xt1 + lam * xt2
        if t == 0:
            return img
        times_all, coefs_all = self._ddpm_tables()  # row i is step T-1-i
        first = self.num_timesteps - t             # the row of step t - 1
        times, coefs = times_all[first:], coefs_all[first:]
        return self._run(DDPM, tuple(x1.shape), times, coefs, [ti > 0 for ti in times], False, noise, seed, text_emb=text_emb,
                         cond=cond, x_init=img, unnormalize=False)

    # -- the elementwise helpers and model_predictions / p_mean_variance as callable methods ------------------------
    def _bt(self, t, b: int) -> torch.Tensor:
        """(B,) CPU long timesteps from an int or a tensor."""
        if isinstance(t, torch.Tensor):
            t = t.detach().to("cpu", torch.long).reshape(-1)
            return t.expand(b).contiguous() if t.numel() == 1 else t
        return torch.full((b,), int(t), dtype=torch.long)

    def _lincomb(self, x, y, name0, name1, t, mode=0, clamp=False, neg1=False):
        """mode 0: extract(name0, t) * x (+|-) extract(name1, t) * y;  mode 1: (extract(name0, t) * x - y) / extract(name1, t)."""
        x = x.to(self.device, torch.float32).contiguous()
        y = y.to(self.device, torch.float32).contiguous()
        assert x.shape == y.shape
        bt = self._bt(t, x.shape[0])
        c0, c1 = self._sched[name0][bt], self._sched[name1][bt]
        coef = torch.stack([c0, -c1 if neg1 else c1], dim=1).to(torch.float32).contiguous()
        out = torch.empty_like(x)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_op_lincomb(_lib.ptr(x), _lib.ptr(y), C.cast(coef.data_ptr(), C.POINTER(C.c_float)), _lib.ptr(out),
                                           x.shape[0], x[0].numel(), mode, int(bool(clamp)), stream))
        return out

    def _ext(self, name, t, x):
        """``extract(buffer, t, x.shape)`` (:394-397) on the device."""
        bt = self._bt(t, x.shape[0])
        return self._sched[name][bt].reshape(-1, *((1,) * (x.dim() - 1))).to(self.device)

    def predict_start_from_noise(self, x_t, t, noise):
        """:570-574."""
        return self._lincomb(x_t, noise, "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", t, neg1=True)

    def predict_noise_from_start(self, x_t, t, x0):
        """:576-580."""
        return self._lincomb(x_t, x0, "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", t, mode=1)

    def predict_v(self, x_start, t, noise):
        """:582-586."""
        return self._lincomb(noise, x_start, "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", t, neg1=True)

    def predict_start_from_v(self, x_t, t, v):
        """:588-592."""
        return self._lincomb(x_t, v, "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", t, neg1=True)

    def q_posterior(self, x_start, x_t, t):
        """:594-601: (posterior_mean, posterior_variance, posterior_log_variance_clipped)."""
        mean = self._lincomb(x_start, x_t, "posterior_mean_coef1", "posterior_mean_coef2", t)
        return mean, self._ext("posterior_variance", t, x_t), self._ext("posterior_log_variance_clipped", t, x_t)

    @torch.inference_mode()
    def model_predictions(self, x, t, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False, **cond_kw):
        """:603-626: ``ModelPrediction(pred_noise, pred_x_start)`` for per-sample timesteps ``t`` (a (B,) tensor, or an int)."""
        x = x.to(self.device, torch.float32).contiguous()
        bt = self._bt(t, x.shape[0])
        if self.self_condition:
            cond_kw = dict(cond_kw, x_self_cond=x_self_cond)
        else:
            assert x_self_cond is None, "the model was built without self_condition"
        out = self._eps(x, bt.to(self.device), **cond_kw)
        clamp = bool(clip_x_start)
        if self.objective == "pred_noise":
            pred_noise = out
            x_start = self._lincomb(x, out, "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", bt, clamp=clamp, neg1=True)
            if clip_x_start and rederive_pred_noise:
                pred_noise = self.predict_noise_from_start(x, bt, x_start)
        elif self.objective == "pred_x0":
            x_start = out.clamp(-1.0, 1.0) if clamp else out
            pred_noise = self.predict_noise_from_start(x, bt, x_start)
        else:
            x_start = self._lincomb(x, out, "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", bt, clamp=clamp, neg1=True)
            pred_noise = self.predict_noise_from_start(x, bt, x_start)
        return ModelPrediction(pred_noise, x_start)

    @torch.inference_mode()
    def p_mean_variance(self, x, t, x_self_cond=None, clip_denoised=True, **cond_kw):
        """:628-636: (model_mean, posterior_variance, posterior_log_variance, x_start)."""
        x = x.to(self.device, torch.float32).contiguous()
        x_start = self.model_predictions(x, t, x_self_cond, clip_x_start=bool(clip_denoised), **cond_kw).pred_x_start
        mean, var, logvar = self.q_posterior(x_start, x, t)
        return mean, var, logvar, x_start

    @torch.inference_mode()
    def ddim_sample_guided(self, shape, sampling_timesteps=None, guide=None, mask=None, clip_denoised=True, *, noise=None,
                           seed=None):
        """:711-781: the DDIM loop with the raw model output as pred_noise (x_start clipped, the noise NOT re-derived) and,
        after every update, ``img = img * mask + q_sample(guide, time) * (1 - mask)``.  Eager (one U-Net forward and four
        elementwise kernels per step).  The reference also opens a matplotlib figure per step (:760-777); that side effect
        is not reproduced.  ``noise`` draws in the reference's order: x_T, then per step the update's noise and, with a
        guide, q_sample's."""
        if sampling_timesteps is None:
            sampling_timesteps = self.sampling_timesteps
        shape = tuple(int(v) for v in shape)
        b = shape[0]
        if seed is None:
            seed = _default_seed()
        draws = [0]

        def draw():
            draws[0] += 1
            return (noise(shape).to(self.device, torch.float32).contiguous() if noise is not None
                    else self._randn(shape, seed, (1 << 21) + draws[0]))

        ac = self._sched["alphas_cumprod"]
        times = torch.linspace(-1, self.num_timesteps - 1, steps=sampling_timesteps + 1)
        times = list(reversed(times.int().tolist()))
        img = draw()
        if guide is not None:
            guide = guide.to(self.device, torch.float32).contiguous()
            mask_full = mask.to(self.device, torch.float32).expand(shape).contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        x_start = None
        for time, time_next in zip(times[:-1], times[1:]):
            pred_noise, x_start = self.model_predictions(img, time, x_start if self.self_condition else None,
                                                         clip_x_start=clip_denoised)
            if time_next < 0:
                img = x_start
                continue
            alpha, alpha_next = ac[time], ac[time_next]
            sigma = self.ddim_sampling_eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            z = draw()

            def comb(x, y, c0, c1):
                coef = torch.tensor([[float(c0), float(c1)]] * b, dtype=torch.float32).contiguous()
                out = torch.empty_like(x)
                _lib.check(self._lib.dm_op_lincomb(_lib.ptr(x), _lib.ptr(y), C.cast(coef.data_ptr(), C.POINTER(C.c_float)),
                                                   _lib.ptr(out), b, x[0].numel(), 0, 0, stream))
                return out

            img = comb(comb(x_start, pred_noise, alpha_next.sqrt(), c), z, 1.0, sigma)
            if guide is not None:
                guide_t = self.q_sample(guide, torch.full((b,), time, dtype=torch.long), draw())
                out = torch.empty_like(img)
                _lib.check(self._lib.dm_op_mask_mix(_lib.ptr(img), _lib.ptr(guide_t), _lib.ptr(mask_full), _lib.ptr(out),
                                                    img.numel(), stream))
                img = out
        return (img + 1) * 0.5  # unnormalize_to_zero_to_one, whatever auto_normalize says (:779)

    def _eps(self, x, bt, **cond_kw):
        """The U-Net call of ``model_predictions`` (:603-606); subclasses thread their condition through ``cond_kw``."""
        return self.model(x, bt, **cond_kw)

    def _p_sample(self, x, t: int, noise, cond_kw, x_self_cond=None):
        """One reverse step (denoising_diffusion.py:638-645): (pred_img, x_start); the update runs in sampler_update_kernel."""
        b = x.shape[0]
        t = int(t)
        x = x.to(self.device, torch.float32).contiguous()
        bt = torch.full((b,), t, device=self.device, dtype=torch.long)
        if self.self_condition:
            cond_kw = dict(cond_kw, x_self_cond=x_self_cond)
        else:
            assert x_self_cond is None, "the model was built without self_condition"
        eps = self._eps(x, bt, **cond_kw)
        s = self._sched
        stream = torch.cuda.current_stream(self.device).cuda_stream
        coef = (C.c_float * _lib.DM_COEFS)(float(s["sqrt_recip_alphas_cumprod"][t]),
                                           float(s["sqrt_recipm1_alphas_cumprod"][t]),
                                           float(s["posterior_mean_coef1"][t]), float(s["posterior_mean_coef2"][t]),
                                           float((0.5 * s["posterior_log_variance_clipped"][t]).exp()),
                                           1.0 if t > 0 else 0.0, float(s["sqrt_alphas_cumprod"][t]),
                                           float(s["sqrt_one_minus_alphas_cumprod"][t]))
        z = None
        if t > 0:
            z = (noise(x.shape).to(self.device, torch.float32).contiguous() if noise is not None
                 else self._randn(x.shape, _default_seed(), 1))
        out = torch.empty_like(x)
        x_start = torch.empty_like(x)
        _lib.check(self._lib.dm_op_sampler_update(DDPM, self._objective_id, _lib.ptr(x), _lib.ptr(eps), _lib.ptr(z), coef,
                                                  _lib.ptr(out), _lib.ptr(x_start), x.numel(), stream))
        return out, x_start

    @torch.inference_mode()
    def p_sample(self, x, t: int, x_self_cond=None, *, noise=None):
        """:638-645.  Returns (pred_img, x_start)."""
        return self._p_sample(x, t, noise, {}, x_self_cond)


class TextConditionalDenoisingDiffusion(DenoisingDiffusion):
    """Pipes ``text_emb`` through the U-Net at every step
    (denoising_diffusion_text_conditional.py:264-453)."""

    def __init__(self, *, model, embedding_file=None, **kwargs):
        super().__init__(model, **kwargs)
        if embedding_file is not None:
            assert os.path.exists(embedding_file), "Pre-computed caption embeddings file not found."
        self.embedding_file = Path(embedding_file) if embedding_file is not None else None

    def get_random_text_condition(self, batch, device):
        """Random caption embeddings from the pickle the training pipeline wrote (:320-363)."""
        assert self.embedding_file is not None, "no embedding_file given; pass text_emb= to sample()"
        with open(self.embedding_file, "rb") as f:
            # The caller's OWN file, written by their embedding-precompute step, exactly as the reference reads it
            # (:331-332).  pickle executes what it loads: never point embedding_file at a file of unknown origin.
            table = pickle.load(f)
        keys = list(table.keys())
        embs, texts = [], []
        for key in random.choices(keys, k=batch):
            data = table[key]
            i = random.randint(0, data["embeddings"].shape[0] - 1)
            embs.append(torch.tensor(data["embeddings"][i], dtype=torch.float))
            texts.append(data["captions"][i])
        return torch.stack(embs, dim=0).to(device), texts

    def _text(self, batch, save_path_for_text, text_emb):
        if text_emb is None:
            text_emb, texts = self.get_random_text_condition(batch, self.device)
            if save_path_for_text is not None:
                mode = "a" if os.path.exists(save_path_for_text) else "w"
                with open(save_path_for_text, mode) as f:
                    for t in texts:
                        f.write(t + "\n")
        return text_emb

    @torch.inference_mode()
    def p_sample_loop(self, shape, save_path_for_text=None, return_all_timesteps=False, *, text_emb=None, **kw):
        text_emb = self._text(shape[0], save_path_for_text, text_emb)
        return super().p_sample_loop(shape, return_all_timesteps, text_emb=text_emb, **kw)

    @torch.inference_mode()
    def ddim_sample(self, shape, save_path_for_text=None, sampling_timesteps=None, return_all_timesteps=False, *,
                    text_emb=None, **kw):
        text_emb = self._text(shape[0], save_path_for_text, text_emb)
        return super().ddim_sample(shape, sampling_timesteps, return_all_timesteps, text_emb=text_emb, **kw)

    @torch.inference_mode()
    def sample(self, batch_size=16, save_path_for_text=None, return_all_timesteps=False, **kw):
        (h, w), channels = self.image_size, self.channels
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        return sample_fn((batch_size, channels, h, w), save_path_for_text, return_all_timesteps=return_all_timesteps,
                         **kw)

    def p_losses(self, x_start, t, text_emb=None, noise=None, offset_noise_strength=None, **kw):
        """denoising_diffusion_text_conditional.py:476-542 (the reference's positional order: x_start, t, text_emb, noise)."""
        return super().p_losses(x_start, t, noise, offset_noise_strength, text_emb=text_emb, **kw)

    def forward(self, img, text_emb=None, *args, **kwargs):
        """denoising_diffusion_text_conditional.py:544-550: the training loss with the caption embeddings."""
        b, c, h, w = img.shape
        assert (h, w) == tuple(self.image_size), f"height and width of image must be {self.image_size}"
        t = torch.randint(0, self.num_timesteps, (b,)).long()
        return self.p_losses(self.normalize(img.to(self.device, torch.float32)), t, text_emb, *args, **kwargs)

    __call__ = forward

    @torch.inference_mode()
    def interpolate(self, x1, x2, t=None, text_emb=None, lam=0.5, *, noise=None, seed=None):
        """denoising_diffusion_text_conditional.py:456-473 (positional order: x1, x2, t, text_emb, lam)."""
        return self._interpolate(x1, x2, t, lam, noise, seed, text_emb=text_emb)

    @torch.inference_mode()
    def model_predictions(self, x, t, text_emb=None, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False):
        """denoising_diffusion_text_conditional.py:274-297 (positional order: x, t, text_emb, x_self_cond)."""
        kw = {"text_emb": text_emb} if text_emb is not None else {}
        return super().model_predictions(x, t, x_self_cond, clip_x_start, rederive_pred_noise, **kw)

    @torch.inference_mode()
    def p_mean_variance(self, x, t, text_emb=None, x_self_cond=None, clip_denoised=True):
        """:299-307."""
        kw = {"text_emb": text_emb} if text_emb is not None else {}
        return super().p_mean_variance(x, t, x_self_cond, clip_denoised, **kw)

    @torch.inference_mode()
    def p_sample(self, x, t: int, text_emb=None, x_self_cond=None, *, noise=None):
        """denoising_diffusion_text_conditional.py:310-317 (the reference's positional order: x, t, text_emb)."""
        return self._p_sample(x, t, noise, {"text_emb": text_emb} if text_emb is not None else {}, x_self_cond)


class ImageConditionalDenoisingDiffusion(DenoisingDiffusion):
    """``ImageConditionalDenoisingDiffusion`` (denoising_diffusion_image_conditional.py:62-229): the condition image
    is concatenated behind x in front of ``init_conv`` at every step (``Unet(cond_channels=...)``, :42-55).

    ``cond=`` (B, cond_channels, H, W; the reference feeds ToTensor() images in [0, 1]) may be passed to every sampling method; without it the condition
    batch is drawn from ``condition_data_folder`` as the reference does (:123-153, needs PIL + torchvision).  The
    reference's ``sample()`` only works for the DDPM loop (its DDIM call passes ``return_condition_image`` into the
    ``sampling_timesteps`` slot and never fetches a condition, :182,225-229); here both samplers take the condition."""

    def __init__(self, *args, condition_data_folder=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.condition_data_folder = condition_data_folder
        assert getattr(self.model.cfg, "cond_channels", 0) > 0, "the model was built without cond_channels"

    def get_random_condition(self, batch, device):
        """:123-153: `batch` random images of the folder, resized / centre-cropped to image_size."""
        if self.condition_data_folder is None:
            raise RuntimeError("no condition_data_folder given; pass cond= to the sampling call")
        from pathlib import Path

        from PIL import Image
        from torchvision import transforms as T

        tf = T.Compose([T.Resize(self.image_size), T.CenterCrop(self.image_size), T.ToTensor()])  # [0, 1], as :130-136
        paths = random.choices(list(Path(self.condition_data_folder).glob("*.*")), k=batch)
        return torch.stack([tf(Image.open(p).convert("RGB")) for p in paths], dim=0).to(device)

    def _cond(self, batch, cond):
        return cond if cond is not None else self.get_random_condition(batch, self.device)

    @torch.inference_mode()
    def p_sample_loop(self, shape, return_condition_image=False, return_all_timesteps=False, *, cond=None, noise=None,
                      seed=None, max_steps=None, sample_offset=0):
        # the reference draws x_T first, then the condition (:159-163); the order only matters for its global RNG
        times, coefs = self._ddpm_tables()
        cond = self._cond(shape[0], cond)
        ret = self._run(DDPM, shape, times, coefs, [t > 0 for t in times], return_all_timesteps, noise, seed, None,
                        max_steps, cond=cond, sample_offset=sample_offset)
        return (cond, ret) if return_condition_image else ret

    @torch.inference_mode()
    def ddim_sample(self, shape, sampling_timesteps=None, cond=None, return_all_timesteps=False, *, noise=None,
                    seed=None, max_steps=None, sample_offset=0):
        if sampling_timesteps is None:
            sampling_timesteps = self.sampling_timesteps
        times, coefs = self._ddim_tables(sampling_timesteps)
        cond = self._cond(shape[0], cond)
        return self._run(DDIM, shape, times, coefs, [bool(c[5] != 0) for c in coefs], return_all_timesteps, noise, seed,
                         None, max_steps, cond=cond, sample_offset=sample_offset)

    @torch.inference_mode()
    def sample(self, batch_size=16, return_condition_image=False, return_all_timesteps=False, *, cond=None, **kw):
        (h, w), channels = self.image_size, self.channels
        shape = (batch_size, channels, h, w)
        if not self.is_ddim_sampling:
            return self.p_sample_loop(shape, return_condition_image, return_all_timesteps, cond=cond, **kw)
        cond = self._cond(batch_size, cond)
        ret = self.ddim_sample(shape, None, cond, return_all_timesteps, **kw)
        return (cond, ret) if return_condition_image else ret

    def forward(self, img, cond=None, *args, **kwargs):
        """denoising_diffusion_image_conditional.py:313-319: the training loss with the condition image."""
        b, c, h, w = img.shape
        assert (h, w) == tuple(self.image_size), f"height and width of image must be {self.image_size}"
        t = torch.randint(0, self.num_timesteps, (b,)).long()
        return self.p_losses(self.normalize(img.to(self.device, torch.float32)), t, *args, cond=cond, **kwargs)

    __call__ = forward

    @torch.inference_mode()
    def interpolate(self, x1, x2, t=None, cond=None, lam=0.5, *, noise=None, seed=None):
        """denoising_diffusion_image_conditional.py:232-249 (positional order: x1, x2, t, cond, lam)."""
        assert cond is not None, "the image-conditional U-Net needs cond="
        return self._interpolate(x1, x2, t, lam, noise, seed, cond=self._cond(x1.shape[0], cond))

    def model_predictions(self, x, t, cond=None, x_self_cond=None, clip_x_start=False, rederive_pred_noise=False):
        """denoising_diffusion_image_conditional.py:78-102 (positional order: x, t, cond, x_self_cond)."""
        assert cond is not None, "the image-conditional U-Net needs cond="
        return super().model_predictions(x, t, x_self_cond, clip_x_start, rederive_pred_noise, cond=cond)

    def p_mean_variance(self, x, t, cond=None, x_self_cond=None, clip_denoised=True):
        """:104-112."""
        assert cond is not None, "the image-conditional U-Net needs cond="
        return super().p_mean_variance(x, t, x_self_cond, clip_denoised, cond=cond)

    @torch.inference_mode()
    def p_sample(self, x, t: int, cond=None, x_self_cond=None, *, noise=None):
        """denoising_diffusion_image_conditional.py:114-120: ``cond`` is what the U-Net sees behind x."""
        assert cond is not None, "the image-conditional U-Net needs cond="
        return self._p_sample(x, t, noise, {"cond": cond}, x_self_cond)


class LatentDiffusion(DenoisingDiffusion):
    """Diffusion on VAE latents, then decode (latent_diffusion.py:9-66)."""

    def __init__(self, model, vae, latent_shape, **kwargs):
        kwargs.setdefault("auto_normalize", True)
        super().__init__(model, image_size=latent_shape[1], **kwargs)
        self.vae = vae
        self.latent_channels = latent_shape[0]
        self.model.channels = self.latent_channels
        self.normalize = _identity  # latent_diffusion.py:25-26
        self.unnormalize = _identity
        self._unnormalize_flag = 0

    def decode(self, latents):
        return self.vae.decode(latents)

    def encode(self, images):
        """latent_diffusion.py:33-40: the (frozen) VAE's latents of a batch of images."""
        latents = self.vae.encode(images)
        return latents[0] if isinstance(latents, tuple) else latents

    def forward(self, real_images, *args, **kwargs):
        """latent_diffusion.py:51-56: the training loss on the VAE latents of the images."""
        return super().forward(self.encode(real_images), *args, **kwargs)

    __call__ = forward

    @torch.inference_mode()
    def sample(self, batch_size=16, return_all_timesteps=False, **kw):
        (h, w), channels = self.image_size, self.channels
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        latents = sample_fn((batch_size, channels, h, w), return_all_timesteps=return_all_timesteps, **kw)
        return self.decode(latents)


class TextConditionalLatentDiffusion(TextConditionalDenoisingDiffusion):
    """``TextConditionalLatentDiffusion`` (latent-diffusion/ldm/models/latent_diffusion_text_conditional.py:11-99): the
    text-conditional loop on VAE latents (normalize / unnormalize are the identity, :36-37), then ``vae.decode``
    (:78-99)."""

    def __init__(self, model, vae, latent_shape, text_emb_dim=512, **kwargs):
        super().__init__(model=model, image_size=latent_shape[1], **kwargs)
        self.vae = vae
        self.latent_channels = latent_shape[0]
        self.model.channels = self.latent_channels
        self.text_emb_dim = text_emb_dim
        self.normalize = _identity
        self.unnormalize = _identity
        self._unnormalize_flag = 0

    def encode(self, images):
        latents = self.vae.encode(images)
        return latents[0] if isinstance(latents, tuple) else latents

    def decode(self, latents):
        return self.vae.decode(latents)

    def forward(self, target, text_emb, *args, **kwargs):
        """latent_diffusion_text_conditional.py:62-76: the text-conditional training loss on the VAE latents of `target`."""
        return super().forward(self.encode(target), text_emb, *args, **kwargs)

    __call__ = forward

    @torch.inference_mode()
    def sample(self, batch_size=16, save_path_for_text=None, return_all_timesteps=False, **kw):
        (h, w), channels = self.image_size, self.channels
        sample_fn = self.p_sample_loop if not self.is_ddim_sampling else self.ddim_sample
        latents = sample_fn((batch_size, channels, h, w), save_path_for_text,
                            return_all_timesteps=return_all_timesteps, **kw)
        return self.decode(latents)


class ImageConditionalLatentDiffusion(ImageConditionalDenoisingDiffusion):
    """``ImageConditionalLatentDiffusion`` (latent-diffusion/ldm/models/latent_diffusion_image_conditional.py:16-166):
    the image-conditional loop on VAE latents.  The condition image is encoded by the (condition) VQ model and rides
    through every step concatenated behind the latent; the result is decoded at the end.  The reference re-encodes the
    (loop-invariant) condition inside every one of its T iterations (:127); here it is encoded once -- the same value.
    normalize / unnormalize are the identity (:43-44)."""

    def __init__(self, model, vae, latent_shape, init_image_size, cond_vae=None, **kwargs):
        super().__init__(model, image_size=tuple(latent_shape[1:]), **kwargs)
        # latents are not images: the reference forces the identity whatever auto_normalize says (:43-44)
        self.normalize = _identity
        self.unnormalize = _identity
        self._unnormalize_flag = 0
        self.vae = vae
        self.cond_vae = cond_vae if cond_vae is not None else vae
        self.init_image_size = init_image_size
        self.latent_channels = latent_shape[0]

    def encode(self, images, cond=False):
        latents = (self.cond_vae if cond else self.vae).encode(images)
        return latents[0] if isinstance(latents, tuple) else latents

    def decode(self, latents, cond=False):
        return (self.cond_vae if cond else self.vae).decode(latents)

    def forward(self, target, cond, *args, **kwargs):
        """latent_diffusion_image_conditional.py:171-183: the training loss on the latents of `target`, conditioned on the
        latents of `cond` (each through its own VQ model)."""
        return super().forward(self.encode(target), *args, cond=self.encode(cond, cond=True), **kwargs)

    __call__ = forward

    def get_random_condition(self, batch, device):
        """:82-111: like the pixel-space variant, at ``init_image_size``."""
        size, self.image_size = self.image_size, (
            self.init_image_size if isinstance(self.init_image_size, (tuple, list)) else (self.init_image_size,) * 2)
        try:
            return super().get_random_condition(batch, device)
        finally:
            self.image_size = size

    @torch.inference_mode()
    def p_sample_loop(self, shape, return_condition_image=False, return_all_timesteps=False, *, cond=None, **kw):
        cond = self._cond(shape[0], cond)
        ret = super().p_sample_loop(shape, False, return_all_timesteps, cond=self.encode(cond, cond=True), **kw)
        return (cond, ret) if return_condition_image else ret

    @torch.inference_mode()
    def ddim_sample(self, shape, sampling_timesteps=None, cond=None, return_all_timesteps=False, **kw):
        """``cond`` is the condition IMAGE (encoded here), unlike the parent's method which takes what the U-Net sees."""
        cond = self._cond(shape[0], cond)
        return super().ddim_sample(shape, sampling_timesteps, self.encode(cond, cond=True), return_all_timesteps, **kw)

    @torch.inference_mode()
    def sample(self, batch_size=16, return_condition_image=False, return_all_timesteps=False, *, cond=None, **kw):
        (h, w), channels = self.image_size, self.channels
        shape = (batch_size, channels, h, w)
        cond = self._cond(batch_size, cond)
        if not self.is_ddim_sampling:
            lat = self.p_sample_loop(shape, False, return_all_timesteps, cond=cond, **kw)
        else:
            lat = self.ddim_sample(shape, None, cond, return_all_timesteps, **kw)
        img = self.decode(lat)
        return (cond, img) if return_condition_image else img

