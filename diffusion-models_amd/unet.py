"""Host-side mirror of the reference ``Unet`` (and its text / image-conditional
subclasses) in front of the HIP library.

Same constructor arguments, ``forward`` signature, ``state_dict`` key names and
error behaviour as
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion.py:233-390
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion_text_conditional.py:86-214
  denoising-diffusion-pytorch/denoising_diffusion/denoising_diffusion_image_conditional.py:31-55
but no torch compute: torch tensors are only the owners of device memory.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from .spec import UnetConfig, unet_param_spec


def _device_index(device) -> int:
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"the HIP path needs a GPU device, got {d}; there is no CPU fallback")
    return d.index if d.index is not None else torch.cuda.current_device()


class Unet:
    """``Unet(dim, dim_mults=..., channels=...)`` -- drop-in for the reference class.

    Weights are supplied with :meth:`load_state_dict` (reference key names).  The
    object is callable: ``eps = unet(x, time)``.
    """

    def __init__(
        self,
        dim,
        init_dim=None,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=3,
        self_condition=False,
        learned_variance=False,
        learned_sinusoidal_cond=False,
        random_fourier_features=False,
        learned_sinusoidal_dim=16,
        sinusoidal_pos_emb_theta=10000,
        dropout=0.0,
        attn_dim_head=32,
        attn_heads=4,
        full_attn=None,
        flash_attn=False,
        text_condition=False,
        text_emb_dim=512,
        use_cross_attn=False,
        cond_channels=0,
        device="cuda:0",
    ):
        # learned_sinusoidal_cond / random_fourier_features (denoising_diffusion.py:86-101, :271-273): forward() only --
        # DenoisingDiffusion asserts such a U-Net away (:456-457), and so does ours
        # attn_heads / attn_dim_head: one value or one per stage (cast_tuple, :294-295).  The attention kernels are specialised
        # for 32-wide heads (the reference's default everywhere it builds a Unet); the head COUNT is per layer.
        if isinstance(attn_dim_head, (tuple, list)):
            if any(int(v) != int(attn_dim_head[0]) for v in attn_dim_head) or len(attn_dim_head) != len(tuple(dim_mults)):
                raise NotImplementedError("per-stage attn_dim_head: the HIP attention kernels take one head width (32)")
            attn_dim_head = int(attn_dim_head[0])
        if isinstance(attn_heads, (tuple, list)):
            attn_heads = tuple(int(v) for v in attn_heads)
            if len(attn_heads) != len(tuple(dim_mults)):
                raise ValueError("attn_heads needs one entry per stage")
        self.cfg = UnetConfig(
            dim=dim, init_dim=init_dim, out_dim=out_dim, dim_mults=tuple(dim_mults), channels=channels,
            self_condition=self_condition, learned_variance=learned_variance,
            sinusoidal_pos_emb_theta=float(sinusoidal_pos_emb_theta), learned_sinusoidal_cond=bool(learned_sinusoidal_cond),
            random_fourier_features=bool(random_fourier_features), learned_sinusoidal_dim=int(learned_sinusoidal_dim),
            attn_dim_head=attn_dim_head,
            attn_heads=attn_heads, full_attn=tuple(full_attn) if full_attn else None,
            cond_channels=cond_channels, text_condition=text_condition, use_cross_attn=use_cross_attn,
            text_emb_dim=text_emb_dim,
        )
        cfg = self.cfg
        self.channels = channels
        self.dropout = float(dropout)  # nn.Dropout(p) of every Block: training mode only (eval / sampling: identity)
        self.self_condition = self_condition
        self.out_dim = cfg.out_dim_
        self.text_condition = text_condition
        self.use_cross_attn = use_cross_attn
        self.random_or_learned_sinusoidal_cond = cfg.random_or_learned_sinusoidal_cond
        self.device = torch.device(device)
        self._dev_index = _device_index(device)
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        self._loaded = False
        self._bucketed = False  # data-parallel gradient buckets (grad_buckets)
        self._comm_stream = None

        c = _lib.UnetCfg()
        c.dim, c.init_dim, c.out_dim = cfg.dim, cfg.init_dim or 0, cfg.out_dim_
        c.channels, c.input_channels, c.n_stages = cfg.channels, cfg.input_channels, cfg.num_stages
        for i, m in enumerate(cfg.dim_mults):
            c.dim_mults[i] = m
        for i, f in enumerate(cfg.full_attn_):
            c.full_attn[i] = int(f)
        c.attn_heads, c.attn_dim_head = cfg.attn_heads_[0], cfg.attn_dim_head
        for i, hd in enumerate(cfg.attn_heads_):
            c.attn_heads_stage[i] = hd
        c.text_mode = 0 if not text_condition else (2 if use_cross_attn else 1)
        c.text_emb_dim = text_emb_dim
        c.sinusoidal_theta = float(sinusoidal_pos_emb_theta)
        c.learned_sinusoidal_dim = cfg.learned_sinusoidal_dim if cfg.random_or_learned_sinusoidal_cond else 0
        _lib.check(self._lib.dm_unet_create(C.byref(c), self._dev_index, C.byref(self._handle)))

    # -- lifecycle -------------------------------------------------------------------------
    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._lib.dm_unet_destroy(h)
            h.value = None

    @property
    def downsample_factor(self) -> int:
        return self.cfg.downsample_factor

    def eval(self):
        return self

    def param_spec(self):
        return unet_param_spec(self.cfg)

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

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Accepts ``Unet.state_dict()`` of the reference (same names, OIHW / [out,in] layouts).

        The first call packs and uploads the weights (``dm_unet_finalize``).  Later calls REFRESH them in place
        (``dm_unet_update_param`` + ``dm_unet_refresh``): only layers whose values changed are re-packed, into the same
        device buffers, so the workspace and the captured step graph survive -- what ``Trainer.train`` needs when it
        samples from the EMA model after every update (denoising_diffusion.py:1190-1198)."""
        spec = dict(self.param_spec())
        set_param = self._lib.dm_unet_update_param if self._loaded else self._lib.dm_unet_set_param
        unexpected = [k for k in state_dict if k not in spec]
        missing = [k for k in spec if k not in state_dict]
        if strict and (unexpected or missing):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        for name, shape in spec.items():
            if name not in state_dict:
                continue
            t = state_dict[name].detach().to(device="cpu", dtype=torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shape)}")
            shp = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(set_param(self._handle, name.encode(), t.data_ptr(), shp, t.dim()))
        if self._loaded:
            _lib.check(self._lib.dm_unet_refresh(self._handle))
        else:
            _lib.check(self._lib.dm_unet_finalize(self._handle))
        self._loaded = True
        return self

    @property
    def workspace_bytes(self) -> int:
        """Bytes of activation workspace the handle owns (tensors are recycled inside a forward)."""
        return int(self._lib.dm_unet_workspace_bytes(self._handle))

    @property
    def graph_captures(self) -> int:
        """How many times a denoise-step graph was captured on this handle (one per sampled shape)."""
        return int(self._lib.dm_unet_graph_captures(self._handle))

    # -- training ----------------------------------------------------
    def train(self, mode: bool = True):
        """``model.train()``: allocate the gradient buffers and pack the input-gradient convolutions (once)."""
        if mode:
            if not self._loaded:
                raise RuntimeError("load_state_dict() must be called before train()")
            if self.self_condition and self.cfg.cond_channels:
                # the reference's own combination is broken: its image-conditional Unet.forward concatenates cond in front
                # of the base forward, whose default x_self_cond = zeros_like(x) then has channels + cond_channels channels
                # against the channels * 2 + cond_channels init_conv expects (denoising_diffusion_image_conditional.py:51-55,
                # denoising_diffusion.py:352-354): every call without a predicted x_self_cond raises
                raise NotImplementedError("self-conditioning with an image condition: the reference's own forward raises "
                                          "whenever x_self_cond is not given (zeros_like of the concatenated input)")
            _lib.check(self._lib.dm_unet_train_enable(self._handle))
            if not getattr(self, "_training", False):
                self.set_dropout_seed(int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()))
            self._training = True
        return self

    def set_dropout_seed(self, seed: int, p: Optional[float] = None):
        """Philox key of the dropout masks of the training step (default: drawn from torch's global generator at the first
        ``train()``); ``p`` overrides the constructor's ``dropout``.  Every loss / backward call draws fresh masks."""
        if p is not None:
            self.dropout = float(p)
        _lib.check(self._lib.dm_unet_train_dropout(self._handle, self.dropout, C.c_uint64(int(seed))))
        return self

    def grad(self, name: str) -> torch.Tensor:
        """Gradient of one parameter (reference name and shape) left by the last loss / backward call."""
        shape = dict(self.param_spec())[name]
        out = torch.empty(tuple(shape), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_unet_get_grad(self._handle, name.encode(), _lib.ptr(out), stream))
        return out

    def grads(self) -> Dict[str, torch.Tensor]:
        return {name: self.grad(name) for name, _ in self.param_spec()}

    def grads_flat(self) -> torch.Tensor:
        """A zero-copy torch VIEW of the library's flat gradient buffer (all parameters, state-dict order): what
        data-parallel training all-reduces in place -- one collective for every gradient."""
        ptr, n = C.c_void_p(), C.c_int64(0)
        _lib.check(self._lib.dm_unet_grads_flat(self._handle, C.byref(ptr), C.byref(n)))

        class _View:  # the CUDA array interface: torch wraps the memory without copying or owning it
            __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f4", "data": (int(ptr.value), False),
                                        "version": 2}

        return torch.as_tensor(_View(), device=self.device)

    def grad_buckets(self, enable: Optional[bool] = None):
        """[(offset, length)] spans (floats) of ``grads_flat()`` in the order the backward pass completes them -- torch DDP's
        gradient buckets.  ``enable=True`` makes every later loss / backward call finish a bucket's weight gradients as soon
        as it has left the bucket's stages (``bucket_wait`` then lets a collective run beside the rest of the pass)."""
        n = self._lib.dm_unet_train_buckets(self._handle, int(bool(enable)) if enable is not None else int(self._bucketed))
        if n < 0:
            raise RuntimeError("grad_buckets() needs a model in training mode")
        if enable is not None:
            self._bucketed = bool(enable)
        spans = []
        for i in range(n):
            off, cnt = C.c_int64(0), C.c_int64(0)
            _lib.check(self._lib.dm_unet_train_bucket(self._handle, i, C.byref(off), C.byref(cnt), 0, None))
            spans.append((int(off.value), int(cnt.value)))
        return spans

    def bucket_wait(self, i: int, stream: "torch.cuda.Stream"):
        """Make ``stream`` wait until bucket ``i`` of the last loss / backward call is complete."""
        off, cnt = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib.dm_unet_train_bucket(self._handle, i, C.byref(off), C.byref(cnt), 1, stream.cuda_stream))

    def optimizer_step(self, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, max_grad_norm=1.0, sync=True):
        """``clip_grad_norm_(max_grad_norm)`` + ``Adam(lr, betas).step()`` of ``Trainer.train``
        (denoising_diffusion.py:1006, :1178-1183) on the device-resident parameters; every packed weight buffer is then
        rebuilt on the device.  Returns the total gradient norm (before clipping): a float, or with ``sync=False`` a
        0-dim device tensor (``clip_grad_norm_`` returns one too) without waiting for the GPU."""
        norm = C.c_float(0.0)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_unet_optimizer_step(self._handle, lr, betas[0], betas[1], eps,
                                                    max_grad_norm if max_grad_norm else 0.0, C.byref(norm) if sync else None,
                                                    stream))
        if sync:
            return float(norm.value)
        out = torch.empty((), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.dm_unet_train_scalar(self._handle, 1, _lib.ptr(out), stream))
        return out

    def ema_update(self, decay: float, copy: bool = False):
        """``ema = ema * decay + online * (1 - decay)`` (or a copy) on the device-resident EMA parameters."""
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_unet_ema_update(self._handle, float(decay), int(bool(copy)), stream))

    def state_dict(self, ema: bool = False) -> Dict[str, torch.Tensor]:
        """The trained parameters (reference names / shapes) from the device-resident state; ``ema=True``: the EMA copy."""
        if not getattr(self, "_training", False):
            if ema:
                raise RuntimeError("state_dict(ema=True) reads the device-resident training state: call train() first")
            if not self._loaded:
                raise RuntimeError("state_dict() before load_state_dict()")
            out = {}
            for name, shape in self.param_spec():  # the handle's host copies
                t = torch.empty(tuple(shape), dtype=torch.float32)
                _lib.check(self._lib.dm_unet_get_param_host(self._handle, name.encode(), t.data_ptr(), t.numel()))
                out[name] = t.to(self.device)
            return out
        return self._train_tensors(1 if ema else 0)

    def named_parameters(self):
        """(name, tensor) in the reference's ``parameters()`` order: copies of the current values (the scripts count them:
        ``sum(p.numel() for p in model.parameters())``, latent-diffusion/train/train_ldm.py:104)."""
        return iter(self.state_dict().items())

    def parameters(self):
        return iter(self.state_dict().values())

    def _train_tensors(self, which: int) -> Dict[str, torch.Tensor]:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        out = {}
        for name, shape in self.param_spec():
            t = torch.empty(tuple(shape), device=self.device, dtype=torch.float32)
            _lib.check(self._lib.dm_unet_get_param(self._handle, name.encode(), which, _lib.ptr(t), stream))
            out[name] = t
        return out

    def _set_train_tensors(self, which: int, tensors: Dict[str, torch.Tensor]):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        keep = []
        for name, shape in self.param_spec():
            t = tensors[name].detach().to(device=self.device, dtype=torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shape)}")
            keep.append(t)
            _lib.check(self._lib.dm_unet_set_train_tensor(self._handle, name.encode(), which, _lib.ptr(t), stream))
        torch.cuda.current_stream(self.device).synchronize()

    def optimizer_state_dict(self, lr=1e-4, betas=(0.9, 0.99), eps=1e-8) -> Dict:
        """``torch.optim.Adam(model.parameters(), lr, betas).state_dict()`` of the device-resident optimiser state -- what
        ``Trainer.save`` stores under ``'opt'`` (denoising_diffusion.py:1006, :1107): per parameter (in ``parameters()``
        order = the state-dict order of the U-Net) ``step`` / ``exp_avg`` / ``exp_avg_sq``; the ``param_groups`` entry is
        taken from a ``torch.optim.Adam`` of the installed torch, so that ``opt.load_state_dict`` accepts it."""
        if not getattr(self, "_training", False):
            raise RuntimeError("optimizer_state_dict() reads the device-resident training state: call train() first")
        names = [n for n, _ in self.param_spec()]
        dummy = [torch.nn.Parameter(torch.empty(0)) for _ in names]
        groups = torch.optim.Adam(dummy, lr=lr, betas=tuple(betas), eps=eps).state_dict()["param_groups"]
        step = int(self._lib.dm_unet_adam_step(self._handle, -1))
        state = {}
        if step > 0:
            m, v = self._train_tensors(2), self._train_tensors(3)
            for i, n in enumerate(names):
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": m[n], "exp_avg_sq": v[n]}
        return {"state": state, "param_groups": groups}

    def load_optimizer_state_dict(self, sd: Dict) -> Dict:
        """Restore ``exp_avg`` / ``exp_avg_sq`` / ``step`` from a ``torch.optim.Adam`` state dict (``Trainer.load``,
        :1125).  Returns the hyper-parameters of the (single) param group (``lr``, ``betas``, ``eps``)."""
        if not getattr(self, "_training", False):
            raise RuntimeError("load_optimizer_state_dict() writes the device-resident training state: call train() first")
        names = [n for n, _ in self.param_spec()]
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(names):
            raise RuntimeError("optimizer state does not match the U-Net: one param group over "
                               f"{len(names)} parameters expected")
        g = groups[0]
        if g.get("weight_decay", 0) or g.get("amsgrad", False) or g.get("maximize", False):
            raise RuntimeError("only the reference's plain Adam (no weight decay / amsgrad / maximize) is on the HIP path")
        state = sd["state"]
        if state:
            idx = g["params"]
            steps = {int(float(state[i]["step"])) for i in idx}
            if len(steps) != 1:
                raise RuntimeError("parameters with different Adam step counts")
            self._set_train_tensors(2, {n: state[i]["exp_avg"] for n, i in zip(names, idx)})
            self._set_train_tensors(3, {n: state[i]["exp_avg_sq"] for n, i in zip(names, idx)})
            self._lib.dm_unet_adam_step(self._handle, steps.pop())
        else:
            self._lib.dm_unet_adam_step(self._handle, 0)
        return {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"]}

    def load_ema_state_dict(self, sd: Dict[str, torch.Tensor]):
        """The EMA copy of the parameters from a dict with the U-Net's names."""
        self._set_train_tensors(1, sd)

    def sync(self):
        """Make the sampling path of THIS handle see the trained weights (device -> host -> re-pack)."""
        _lib.check(self._lib.dm_unet_train_sync(self._handle))
        return self

    def check_device_pack(self) -> int:
        """Self-check of the device-side weight packers against the host packers: number of differing buffers."""
        n = self._lib.dm_unet_check_device_pack(self._handle)
        if n < 0:
            _lib.check(1)
        if n > 0:
            print(self._lib.dm_last_error().decode(errors="replace"))
        return n

    # -- forward ---------------------------------------------------------------------------
    def _ctx(self, text_emb: Optional[torch.Tensor], batch: int):
        if text_emb is None or not self.text_condition:
            return None, 0
        t = text_emb
        if t.dim() == 2:
            t = t.unsqueeze(1)
        if t.shape[0] != batch or t.shape[2] != self.cfg.text_emb_dim:
            raise RuntimeError(f"text_emb shape {tuple(text_emb.shape)} does not match batch / text_emb_dim")
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        return t, t.shape[1]

    def forward(self, x, time, *args, x_self_cond=None, text_emb=None, cond=None):
        """One class stands for the reference's three ``Unet`` classes, whose positional orders differ:
        ``forward(x, time, x_self_cond)`` (denoising_diffusion.py:349), ``forward(x, time, text_emb, x_self_cond)`` for the
        text-conditional one (denoising_diffusion_text_conditional.py:131) and keyword-only ``cond`` / ``x_self_cond`` for
        the image-conditional one (denoising_diffusion_image_conditional.py:51).  Extra positional arguments are read in
        the order of the variant this object was built as."""
        if args:
            order = ("text_emb", "x_self_cond") if self.text_condition else ("x_self_cond",)
            assert self.cfg.cond_channels == 0, "the image-conditional Unet takes cond / x_self_cond by keyword only"
            assert len(args) <= len(order), "too many positional arguments"
            given = dict(x_self_cond=x_self_cond, text_emb=text_emb)
            for name, val in zip(order, args):
                assert given[name] is None, f"{name} given twice"
                given[name] = val
            x_self_cond, text_emb = given["x_self_cond"], given["text_emb"]
        if not self._loaded:
            raise RuntimeError("load_state_dict() must be called before forward()")
        f = self.downsample_factor
        assert all(d % f == 0 for d in x.shape[-2:]), (
            f"your input dimensions {tuple(x.shape[-2:])} need to be divisible by {f}, given the unet"
        )
        x = x.to(device=self.device, dtype=torch.float32)
        if self.self_condition:
            if x_self_cond is None:
                x_self_cond = torch.zeros_like(x)
            x = torch.cat((x_self_cond.to(x), x), dim=1)
        if cond is not None:
            x = torch.cat((x, cond.to(x)), dim=1)
        x = x.contiguous()
        B, Cin, H, W = x.shape
        if Cin != self.cfg.input_channels:
            raise RuntimeError(f"expected {self.cfg.input_channels} input channels, got {Cin}")
        t = time.to(device=self.device, dtype=torch.int64).reshape(-1)
        if t.numel() == 1 and B > 1:
            t = t.expand(B)  # the reference broadcasts a single time embedding over the batch
        if t.numel() != B:
            raise RuntimeError(f"time has {t.numel()} entries for a batch of {B}")
        t = t.contiguous()
        ctx, m = self._ctx(text_emb, B)
        out = torch.empty((B, self.out_dim, H, W), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_unet_forward(self._handle, _lib.ptr(x), _lib.ptr(t), _lib.ptr(ctx), m,
                                             _lib.ptr(out), B, H, W, stream))
        return out

    __call__ = forward
