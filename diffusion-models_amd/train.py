"""Host-side mirror of one iteration of ``Trainer.train`` (denoising-diffusion-pytorch/denoising_diffusion/
denoising_diffusion.py:1162-1190) on the HIP training step: the micro-batch loop with ``loss / gradient_accumulate_every``,
``clip_grad_norm_(max_grad_norm)``, ``Adam.step()``, ``ema.update()`` -- every FLOP in libdm_hip.so, the parameters, Adam
moments and the EMA copy resident on the device.

``EMA`` restates the schedule of ``ema_pytorch.EMA`` as the reference configures it (``EMA(diffusion_model, beta=ema_decay,
update_every=ema_update_every)``, :1017): ema-pytorch is not in the reference tree nor in this image, so the schedule is
restated from its published source (ema-pytorch 0.7.x: ``update_after_step=100``, ``inv_gamma=1``, ``power=2/3``,
``min_value=0``) -- parity unpinned, the arithmetic itself (copy / lerp) is tested against torch.

The reference's ``Trainer`` (datasets, accelerate, checkpoints, tensorboard, FID hooks) is outside the scope of this
package; INTEGRATION.md shows the three lines its loop changes.
"""
from __future__ import annotations

from typing import Iterable, Optional

import torch


class EMA:
    """``ema = EMA(diffusion, beta=0.995, update_every=10); ema.update(); ema.ema_model.sample(...)``."""

    def __init__(self, model, beta=0.9999, update_after_step=100, update_every=10, inv_gamma=1.0, power=2 / 3,
                 min_value=0.0):
        self.online_model = model
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.step = 0
        self.initted = False
        self._ema_model = None
        self._ema_model_step = -1

    def get_current_decay(self) -> float:
        epoch = max(self.step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    def update(self):
        step = self.step
        self.step += 1
        if step % self.update_every != 0:
            return
        unet = self.online_model.model
        if step <= self.update_after_step or not self.initted:
            unet.ema_update(0.0, copy=True)
            self.initted = self.initted or step > self.update_after_step
            return
        unet.ema_update(self.get_current_decay(), copy=False)

    def state_dict(self):
        """``ema_pytorch.EMA.state_dict()`` as ``Trainer.save`` stores it under ``'ema'`` (:1108): the averaged copy under
        ``ema_model.`` (the prefix the reference's sampling scripts strip, denoising-diffusion-pytorch/sampling.py:157-159),
        the online model under ``online_model.``, and the ``initted`` / ``step`` buffers (names from ema-pytorch's
        published source; parity unpinned beyond the ``ema_model.`` prefix)."""
        d = self.online_model
        out = {"initted": torch.tensor(bool(self.initted)), "step": torch.tensor(int(self.step))}
        online = diffusion_state_dict(d)
        out.update({"online_model." + k: v for k, v in online.items()})
        averaged = diffusion_state_dict(d, ema=True) if self.step > 0 else online
        out.update({"ema_model." + k: v for k, v in averaged.items()})
        return out

    def load_state_dict(self, sd):
        """``ema.load_state_dict(data['ema'])`` as ``Trainer.load`` (:1127) and the sampling scripts
        (denoising-diffusion-pytorch/sampling.py:157-159) call it.  On a model in training mode the averaged parameters go
        into the device-resident EMA state; on an inference-only model (the sampling scripts) they go straight into
        ``ema_model``, which is then ready to sample."""
        self.initted = bool(sd["initted"]) if "initted" in sd else True
        self.step = int(sd["step"]) if "step" in sd else 1
        pre = "ema_model.model."
        averaged = {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        if not averaged:
            raise KeyError(f"state dict has no '{pre}*' entries")
        if getattr(self.online_model.model, "_training", False):
            if self.step > 0:
                self.online_model.model.load_ema_state_dict(averaged)
            self._ema_model_step = -1
        else:
            self._make_ema_model().model.load_state_dict(averaged)
            self._ema_model_step = self.step
        return self

    def _make_ema_model(self):
        """A shallow copy of the online diffusion object (same class, schedule, VAE, options) around a second U-Net handle
        of the same variant -- what ``copy.deepcopy(model)`` is to ``ema_pytorch.EMA``."""
        import copy

        from .unet import Unet

        d = self.online_model
        if self._ema_model is None:
            cfg = d.model.cfg
            unet = Unet(dim=cfg.dim, init_dim=cfg.init_dim, out_dim=cfg.out_dim, dim_mults=cfg.dim_mults, channels=cfg.channels,
                        self_condition=cfg.self_condition, sinusoidal_pos_emb_theta=cfg.sinusoidal_pos_emb_theta,
                        attn_dim_head=cfg.attn_dim_head, attn_heads=cfg.attn_heads, full_attn=cfg.full_attn,
                        text_condition=cfg.text_condition, text_emb_dim=cfg.text_emb_dim, use_cross_attn=cfg.use_cross_attn,
                        cond_channels=cfg.cond_channels, device=d.device)
            self._ema_model = copy.copy(d)
            self._ema_model.model = unet
        return self._ema_model

    @property
    def ema_model(self):
        """A sampler holding the EMA weights (a second U-Net handle, refreshed when the EMA state has moved on)."""
        d = self.online_model
        self._make_ema_model()
        if self._ema_model_step != self.step:
            if self.step == 0:
                raise RuntimeError("EMA.update() has not run yet")
            self._ema_model.model.load_state_dict(d.model.state_dict(ema=True))
            self._ema_model_step = self.step
        return self._ema_model


def diffusion_state_dict(diffusion, ema: bool = False):
    """``DenoisingDiffusion.state_dict()`` of the reference module: the 13 schedule buffers + ``model.*`` (the online
    parameters of the device-resident training state, or its EMA copy)."""
    out = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in diffusion.state_dict().items()}
    if ema:
        out.update({"model." + k: v.cpu() for k, v in diffusion.model.state_dict(ema=True).items()})
    return out


def save_checkpoint(path, diffusion, *, step: int, ema: Optional[EMA] = None, lr=1e-4, betas=(0.9, 0.99), eps=1e-8,
                    version: str = "dm_hip"):
    """``Trainer.save`` (:1100-1113): ``{'step', 'model', 'opt', 'ema', 'scaler', 'version'}`` in the reference's layouts
    (``model``: ``DenoisingDiffusion.state_dict()``; ``opt``: ``torch.optim.Adam.state_dict()``; ``ema``:
    ``ema_pytorch.EMA.state_dict()``), so that the reference's ``Trainer.load`` / sampling scripts -- and
    ``load_checkpoint`` below -- read it back.  Tensors are written on the CPU."""
    opt = diffusion.model.optimizer_state_dict(lr=lr, betas=betas, eps=eps)
    for st in opt["state"].values():
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].cpu(), st["exp_avg_sq"].cpu()
    data = {"step": int(step), "model": diffusion_state_dict(diffusion), "opt": opt,
            "ema": ema.state_dict() if ema is not None else None, "scaler": None, "version": version}
    torch.save(data, str(path))
    return data


def load_checkpoint(path, diffusion, *, ema: Optional[EMA] = None):
    """``Trainer.load`` (:1115-1133) into a diffusion object in training mode: the online parameters, Adam's moments and
    step counter, the EMA copy and its counters.  Returns ``(step, adam_hyper_parameters)``.  The file is opened with
    ``weights_only=True`` (nothing from it is executed)."""
    data = torch.load(str(path), map_location="cpu", weights_only=True)
    diffusion.load_state_dict(data["model"])
    diffusion.train()
    hyper = diffusion.model.load_optimizer_state_dict(data["opt"])
    if ema is not None and data.get("ema") is not None:
        ema.load_state_dict(data["ema"])
    return int(data["step"]), hyper


def train_step(diffusion, micro_batches: Iterable[torch.Tensor], *, lr=1e-4, betas=(0.9, 0.99), eps=1e-8,
               max_grad_norm=1.0, ema: Optional[EMA] = None, t=None, noise=None, group=None, sync=True,
               timing: Optional[dict] = None, bucketed: Optional[bool] = None):
    """One iteration of ``Trainer.train`` (:1164-1190).  ``micro_batches``: the ``gradient_accumulate_every`` image batches
    (in [0, 1]) of the iteration.  ``t`` / ``noise`` (lists, one per micro-batch) inject the random draws for tests.
    Under ``torch.distributed`` (one process per GPU, as ``accelerate`` runs the reference's Trainer under DDP) every rank
    computes the gradients of ITS micro-batches and the flat gradient buffer is averaged over the ranks in place (RCCL over
    xGMI) before the optimiser step; every rank then takes the same step.  With more than one rank (``bucketed=True`` forces
    it, ``False`` forbids it) the buffer is all-reduced BUCKET BY BUCKET on a second stream, each bucket as soon as the
    backward pass has completed it -- DDP's overlap of communication with the backward pass (``Unet.grad_buckets``);
    otherwise as one collective.
    Returns (total_loss of this rank, grad_norm): floats, or with ``sync=False`` 0-dim device tensors -- the iteration is
    then only ENQUEUED when the call returns (no host round trip: the reference's loop runs ahead of the GPU the same way
    until ``loss.item()``), so back-to-back iterations leave no idle gaps on the GPU.  ``timing`` (a dict): HIP event pairs
    around the gradient all-reduce are appended to ``timing["allreduce_events"]`` (bench.py reports its share)."""
    import torch.distributed as dist

    batches = list(micro_batches)
    k = len(batches)
    total = 0.0
    lazy = {} if sync else {"sync": False}  # only the asynchronous form asks the diffusion object for anything new
    parallel = dist.is_available() and dist.is_initialized()  # also at world size 1: the collectives are the same code path
    unet = diffusion.model
    # by default on RCCL with more than one rank (gloo stages every collective through pinned host memory: nothing to overlap)
    overlap = parallel and bucketed is not False and hasattr(unet, "grad_buckets") and (
        bucketed or (dist.get_world_size(group) > 1 and dist.get_backend(group) == "nccl"))
    if overlap != getattr(unet, "_bucketed", False) and hasattr(unet, "grad_buckets"):
        unet.grad_buckets(enable=overlap)
    for i, data in enumerate(batches):
        x = diffusion.normalize(data.to(diffusion.device, torch.float32))
        ti = t[i] if t is not None else torch.randint(0, diffusion.num_timesteps, (x.shape[0],)).long()
        ni = noise[i] if noise is not None else None
        loss = diffusion.p_losses(x, ti, noise=ni, loss_scale=1.0 / k, accumulate=i > 0, **lazy)
        total = (total + float(loss)) if sync else (loss if i == 0 else total + loss)
    if parallel:
        # DDP averages the gradients over the ranks.  The buffer is the library's own (a zero-copy view); no host
        # synchronisation anywhere: each collective is ordered behind the kernels that complete its span, and the optimiser
        # step behind the collectives.
        flat, world = unet.grads_flat(), dist.get_world_size(group)
        ev = None
        if timing is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if overlap:
            # bucket by bucket, on a second stream: a bucket's all-reduce starts when the backward pass (still running --
            # the call above only enqueued it) has completed that bucket, and runs beside the rest of the pass
            main = torch.cuda.current_stream(diffusion.device)
            if unet._comm_stream is None:
                unet._comm_stream = torch.cuda.Stream(device=diffusion.device)
            side = unet._comm_stream
            for b, (off, n) in enumerate(unet.grad_buckets()):
                unet.bucket_wait(b, side)
                with torch.cuda.stream(side):
                    span = flat[off:off + n]
                    dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group)
                    span.div_(world)
            main.wait_stream(side)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            flat.div_(world)
        if ev is not None:
            ev[1].record()
            timing.setdefault("allreduce_events", []).append(ev)
    norm = diffusion.model.optimizer_step(lr=lr, betas=betas, eps=eps, max_grad_norm=max_grad_norm, **lazy)
    if ema is not None:
        ema.update()
    return total, norm
