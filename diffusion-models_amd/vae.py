"""Host-side mirror of ``VQModel.decode`` (latent-diffusion/ldm/models/autoencoder.py:113-116):
``post_quant_conv`` -> ``Decoder`` (latent-diffusion/ldm/modules/diffusionmodules/model.py:476-585),
executed by ``dm_decoder_forward`` in libdm_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib
from .spec import DecoderConfig, EncoderConfig, decoder_param_spec, encoder_param_spec
from .unet import _device_index


class VQDecoder:
    """``vae = VQDecoder(ddconfig, embed_dim); vae.load_state_dict(vqmodel_state_dict); vae.decode(z)``.

    ``load_state_dict`` takes the ``state_dict`` of the reference ``VQModel`` (or a Lightning
    checkpoint's ``state_dict``) and uses the ``post_quant_conv.*`` and ``decoder.*`` entries."""

    def __init__(self, ddconfig: Dict, embed_dim: int, device="cuda:0"):
        self.cfg = DecoderConfig(
            ch=ddconfig["ch"], out_ch=ddconfig["out_ch"], ch_mult=tuple(ddconfig["ch_mult"]),
            num_res_blocks=ddconfig["num_res_blocks"], attn_resolutions=tuple(ddconfig.get("attn_resolutions", ())),
            resolution=ddconfig["resolution"], z_channels=ddconfig["z_channels"], embed_dim=embed_dim,
        )
        cfg = self.cfg
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        self._loaded = False
        c = _lib.DecoderCfg()
        c.ch, c.out_ch, c.n_levels = cfg.ch, cfg.out_ch, cfg.num_resolutions
        for i, m in enumerate(cfg.ch_mult):
            c.ch_mult[i] = m
        c.num_res_blocks = cfg.num_res_blocks
        c.n_attn_res = len(cfg.attn_resolutions)
        for i, r in enumerate(cfg.attn_resolutions):
            c.attn_resolutions[i] = r
        c.resolution, c.z_channels, c.embed_dim = cfg.resolution, cfg.z_channels, cfg.embed_dim
        _lib.check(self._lib.dm_decoder_create(C.byref(c), _device_index(device), C.byref(self._handle)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._lib.dm_decoder_destroy(h)
            h.value = None

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def param_spec(self):
        return decoder_param_spec(self.cfg)

    def load_state_dict(self, state_dict, strict: bool = True):
        spec = dict(self.param_spec())
        missing = [k for k in spec if k not in state_dict]
        if missing:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}")
        for name, shape in spec.items():
            t = state_dict[name].detach().to(device="cpu", dtype=torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shape)}")
            shp = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self._lib.dm_decoder_set_param(self._handle, name.encode(), t.data_ptr(), shp, t.dim()))
        _lib.check(self._lib.dm_decoder_finalize(self._handle))
        self._loaded = True
        return self

    def decoded_shape(self, latent_chw):
        """(C, H, W) of ``decode`` for one latent of shape (embed_dim, h, w)."""
        f = 2 ** (self.cfg.num_resolutions - 1)
        return (self.cfg.out_ch, int(latent_chw[1]) * f, int(latent_chw[2]) * f)

    @torch.inference_mode()
    def decode(self, quant):
        if not self._loaded:
            raise RuntimeError("load_state_dict() must be called before decode()")
        z = quant.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cz, h, w = z.shape
        if Cz != self.cfg.embed_dim:
            raise RuntimeError(f"expected {self.cfg.embed_dim} latent channels, got {Cz}")
        f = 2 ** (self.cfg.num_resolutions - 1)
        out = torch.empty((B, self.cfg.out_ch, h * f, w * f), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_decoder_forward(self._handle, _lib.ptr(z), _lib.ptr(out), B, h, w, stream))
        return out


class VQEncoder:
    """Mirror of ``VQModel.encode`` / ``encode_to_prequant`` (autoencoder.py:102-111): ``Encoder`` -> ``quant_conv`` ->
    nearest-codebook quantisation, executed by ``dm_encoder_forward``.  ``load_state_dict`` takes the ``VQModel``
    state dict and uses its ``encoder.*``, ``quant_conv.*`` and ``quantize.embedding.weight`` entries."""

    def __init__(self, ddconfig: Dict, embed_dim: int, n_embed: int, device="cuda:0"):
        self.cfg = EncoderConfig(
            ch=ddconfig["ch"], in_channels=ddconfig.get("in_channels", 3), ch_mult=tuple(ddconfig["ch_mult"]),
            num_res_blocks=ddconfig["num_res_blocks"], attn_resolutions=tuple(ddconfig.get("attn_resolutions", ())),
            resolution=ddconfig["resolution"], z_channels=ddconfig["z_channels"], embed_dim=embed_dim, n_embed=n_embed,
            double_z=bool(ddconfig.get("double_z", False)),
        )
        cfg = self.cfg
        self.device = torch.device(device)
        self._lib = _lib.load()
        self._handle = C.c_void_p()
        self._loaded = False
        c = _lib.EncoderCfg()
        c.ch, c.in_channels, c.n_levels = cfg.ch, cfg.in_channels, cfg.num_resolutions
        for i, m in enumerate(cfg.ch_mult):
            c.ch_mult[i] = m
        c.num_res_blocks = cfg.num_res_blocks
        c.n_attn_res = len(cfg.attn_resolutions)
        for i, r in enumerate(cfg.attn_resolutions):
            c.attn_resolutions[i] = r
        c.resolution, c.z_channels, c.embed_dim, c.n_embed = cfg.resolution, cfg.z_channels, cfg.embed_dim, cfg.n_embed
        c.double_z = int(cfg.double_z)
        _lib.check(self._lib.dm_encoder_create(C.byref(c), _device_index(device), C.byref(self._handle)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self._lib.dm_encoder_destroy(h)
            h.value = None

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def param_spec(self):
        return encoder_param_spec(self.cfg)

    def load_state_dict(self, state_dict, strict: bool = True):
        spec = dict(self.param_spec())
        missing = [k for k in spec if k not in state_dict]
        if missing:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}")
        for name, shape in spec.items():
            t = state_dict[name].detach().to(device="cpu", dtype=torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shape)}")
            shp = (C.c_int64 * t.dim())(*t.shape)
            _lib.check(self._lib.dm_encoder_set_param(self._handle, name.encode(), t.data_ptr(), shp, t.dim()))
        _lib.check(self._lib.dm_encoder_finalize(self._handle))
        self._loaded = True
        return self

    def _run(self, x, want_quant: bool):
        if not self._loaded:
            raise RuntimeError("load_state_dict() must be called before encode()")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, Cin, H, W = x.shape
        if Cin != self.cfg.in_channels:
            raise RuntimeError(f"expected {self.cfg.in_channels} image channels, got {Cin}")
        f = 2 ** (self.cfg.num_resolutions - 1)
        shape = (B, self.cfg.embed_dim, H // f, W // f)
        pre = torch.empty(shape, device=self.device, dtype=torch.float32)
        zq = torch.empty(shape, device=self.device, dtype=torch.float32) if want_quant else None
        idx = torch.empty((B * shape[2] * shape[3],), device=self.device, dtype=torch.int32) if want_quant else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._lib.dm_encoder_forward(self._handle, _lib.ptr(x), _lib.ptr(zq), _lib.ptr(pre), _lib.ptr(idx), B, H,
                                                W, stream))
        return zq, pre, idx

    @torch.inference_mode()
    def encode(self, x):
        """(quant, emb_loss, (perplexity, min_encodings, indices)) like ``VQModel.encode``; the loss terms are training
        quantities and come back as ``None``."""
        zq, _, idx = self._run(x, True)
        return zq, None, (None, None, idx.to(torch.int64))

    @torch.inference_mode()
    def encode_to_prequant(self, x):
        return self._run(x, False)[1]


class VQModel:
    """``VQModel`` of the reference restricted to inference: ``encode`` / ``encode_to_prequant`` / ``decode`` over one
    ``state_dict`` (autoencoder.py:14-121)."""

    def __init__(self, ddconfig: Dict, lossconfig=None, n_embed: int = None, embed_dim: int = None, ckpt_path=None,
                 ignore_keys=(), image_key="image", colorize_nlabels=None, monitor=None, batch_resize_range=None,
                 scheduler_config=None, lr_g_factor=1.0, remap=None, sane_index_shape=False, use_ema=False, *,
                 device="cuda:0"):
        """The reference's positional order (autoencoder.py:15-31: ddconfig, lossconfig, n_embed, embed_dim, ...).  The loss
        network, the Lightning / EMA / scheduler options are training-only and ignored; ``ckpt_path`` loads the state dict
        (``load_vae_checkpoint``: ``weights_only=True``) minus the keys that start with one of ``ignore_keys``."""
        assert n_embed is not None and embed_dim is not None, "n_embed and embed_dim are required"
        assert remap is None and not sane_index_shape, "remap / sane_index_shape are not on the HIP path"
        self.image_key = image_key
        self.encoder = VQEncoder(ddconfig, embed_dim, n_embed, device=device)
        self.decoder = VQDecoder(ddconfig, embed_dim, device=device)
        self.device = torch.device(device)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    def init_from_ckpt(self, path, ignore_keys=()):
        """autoencoder.py:78-91 through the safe loader of checkpoint.py."""
        from .checkpoint import load_vae_checkpoint

        sd = {k: v for k, v in load_vae_checkpoint(path).items() if not any(k.startswith(ik) for ik in ignore_keys)}
        self.load_state_dict(sd, strict=False)
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def param_spec(self):
        return self.encoder.param_spec() + self.decoder.param_spec()

    def load_state_dict(self, state_dict, strict: bool = True):
        self.encoder.load_state_dict(state_dict, strict)
        self.decoder.load_state_dict(state_dict, strict)
        return self

    def encode(self, x):
        return self.encoder.encode(x)

    def encode_to_prequant(self, x):
        return self.encoder.encode_to_prequant(x)

    def decode(self, quant):
        return self.decoder.decode(quant)

    def forward(self, input, return_pred_indices=False):
        """autoencoder.py:123-128: encode, quantize, decode -> (dec, diff[, indices]); the commitment loss ``diff`` is a
        training quantity and comes back as ``None``."""
        quant, diff, (_, _, ind) = self.encode(input)
        dec = self.decode(quant)
        return (dec, diff, ind) if return_pred_indices else (dec, diff)

    __call__ = forward

    def decoded_shape(self, latent_chw):
        return self.decoder.decoded_shape(latent_chw)

