"""Host-side mirror of ``VQModel.decode`` (latent-diffusion/ldm/models/autoencoder.py:113-116):
``post_quant_conv`` -> ``Decoder`` (latent-diffusion/ldm/modules/diffusionmodules/model.py:476-585),
executed by ``dm_decoder_forward`` in libdm_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import _lib
from .spec import DecoderConfig, decoder_param_spec
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
