"""Host-side mirror of the sample CONSUMER next to the sampling path: InceptionV3 features on
the GPU for the reference's FID and Inception-score evaluators.

Reference surface mirrored (paths relative to the reference checkout):
  denoising-diffusion-pytorch/denoising_diffusion/fid_evaluation.py:15-133       FIDEvaluation
  denoising-diffusion-pytorch/denoising_diffusion/inception_score_evaluation.py:11-113  InceptionScoreEvaluation
The network itself is not in the reference: it comes from pytorch_fid / torchvision (absent here, weights not
fetchable), so its architecture is restated from the published sources (inception_spec.py) -- **parity unpinned**.
The layer graph is this file; every operator runs in libdm_hip.so (``dm_conv_*`` with BatchNorm folded in and ReLU
fused, ``dm_op_pool2d``, ``dm_op_resize_bilinear``, ``dm_op_copy_channels_nhwc``, ``dm_op_global_avgpool``,
``dm_op_linear``).  torch owns the activation buffers (NHWC fp32) and nothing else; there is no CPU fallback.

Without real weights the scores mean nothing: pass ``state_dict=`` (torchvision parameter names, e.g. pytorch_fid's
``pt_inception-2015-12-05`` file or torchvision's ``inception_v3`` checkpoint loaded with ``weights_only=True``).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import warnings
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .inception_spec import BLOCKS, BLOCK_INDEX_BY_DIM, BN_EPS, FEATURE_DIM, STEM, block_convs, inception_param_spec
from .synth import synth_state_dict
from .unet import _device_index

POOL_MAX, POOL_AVG, POOL_AVG_VALID = 0, 1, 2


class _Conv:
    """One BasicConv2d: conv (no bias) + BatchNorm2d(eval, eps 1e-3) folded on the host + ReLU in the kernel epilogue."""

    def __init__(self, lib, sd, spec, dev_index):
        name, cin, cout, (kh, kw), stride, (ph, pw) = spec
        w = sd[name + ".conv.weight"].detach().to("cpu", torch.float64)
        g, b = sd[name + ".bn.weight"].double(), sd[name + ".bn.bias"].double()
        mu, var = sd[name + ".bn.running_mean"].double(), sd[name + ".bn.running_var"].double()
        if tuple(w.shape) != (cout, cin, kh, kw):
            raise RuntimeError(f"size mismatch for {name}.conv.weight: {tuple(w.shape)}")
        s = g / torch.sqrt(var + BN_EPS)
        wf = (w * s.view(-1, 1, 1, 1)).to(torch.float32).contiguous()
        bf = (b - mu * s).to(torch.float32).contiguous()
        self.cout, self.k, self.stride, self.pad = cout, (kh, kw), stride, (ph, pw)
        self._lib = lib
        self._h = C.c_void_p()
        _lib.check(lib.dm_conv_create(wf.data_ptr(), bf.data_ptr(), cout, cin, kh, kw, stride, ph, pw, 1, dev_index,
                                      C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.dm_conv_destroy(h)
            h.value = None

    def out_hw(self, H, W):
        return ((H + 2 * self.pad[0] - self.k[0]) // self.stride + 1, (W + 2 * self.pad[1] - self.k[1]) // self.stride + 1)


class InceptionV3:
    """``InceptionV3(output_blocks=(3,))(x)`` like ``pytorch_fid.inception.InceptionV3`` (the FID graph), or with
    ``variant="torchvision"`` the plain torchvision graph + ``fc`` (``logits(x)``) that the Inception-score evaluator
    calls.  x is (B, 3, H, W) on any device; outputs are NCHW tensors on the GPU."""

    BLOCK_INDEX_BY_DIM = BLOCK_INDEX_BY_DIM
    DEFAULT_BLOCK_INDEX = 3

    def __init__(self, output_blocks: Sequence[int] = (3,), resize_input=True, normalize_input=True, requires_grad=False,
                 variant="fid", state_dict: Optional[Dict[str, torch.Tensor]] = None, device="cuda:0",
                 allow_synthetic=False):
        assert variant in ("fid", "torchvision")
        self.output_blocks = sorted(output_blocks)
        assert max(self.output_blocks) <= 3, "Last possible output block index is 3"
        self.resize_input, self.normalize_input, self.variant = resize_input, normalize_input, variant
        self.device = torch.device(device)
        self._dev_index = _device_index(device)
        self._lib = _lib.load()
        self.synthetic_weights = state_dict is None
        if state_dict is None:
            if not allow_synthetic:
                # the reference fails here too (its constructors download the pretrained weights); returning
                # plausible-looking scores from random weights would be worse
                raise RuntimeError("InceptionV3 needs state_dict= (the pytorch_fid / torchvision pretrained weights, "
                                   "same key names); pass allow_synthetic=True for name-seeded random weights "
                                   "(kernel tests only: FID / IS values computed with them are meaningless)")
            warnings.warn("InceptionV3 without state_dict=: name-seeded SYNTHETIC weights (no network to fetch the "
                          "pretrained ones); FID / IS values computed with them are meaningless")
            state_dict = synth_state_dict(inception_param_spec(), salt=0)
        missing = [k for k, _ in inception_param_spec(with_fc=(variant == "torchvision")) if k not in state_dict]
        if missing:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}")
        self._convs: Dict[str, _Conv] = {}
        for spec in STEM:
            self._convs[spec[0]] = _Conv(self._lib, state_dict, spec, self._dev_index)
        for kind, name, args in BLOCKS:
            for short, spec in block_convs(kind, name, args).items():
                self._convs[f"{name}.{short}"] = _Conv(self._lib, state_dict, spec, self._dev_index)
        self._fc = None
        if variant == "torchvision":
            self._fc = (state_dict["fc.weight"].detach().to(self.device, torch.float32).contiguous(),
                        state_dict["fc.bias"].detach().to(self.device, torch.float32).contiguous())

    def eval(self):
        return self

    def to(self, device):
        assert torch.device(device).type == "cuda", "the HIP path has no CPU fallback"
        return self

    # ---- operators (NHWC tensors owned by torch) ---------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _conv(self, name, x, in_nchw=False):
        c = self._convs[name]
        if in_nchw:
            B, _, H, W = x.shape
        else:
            B, H, W, _ = x.shape
        Ho, Wo = c.out_hw(H, W)
        y = torch.empty((B, Ho, Wo, c.cout), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.dm_conv_forward(c._h, _lib.ptr(x), 1 if in_nchw else 0, B, H, W, _lib.ptr(y), self._stream()))
        return y

    def _pool(self, x, k, stride, pad, mode):
        B, H, W, Cc = x.shape
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((B, Ho, Wo, Cc), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.dm_op_pool2d(_lib.ptr(x), _lib.ptr(y), B, H, W, Cc, k, stride, pad, mode, self._stream()))
        return y

    def _cat(self, parts: List[torch.Tensor]):
        B, H, W, _ = parts[0].shape
        Cd = sum(p.shape[3] for p in parts)
        y = torch.empty((B, H, W, Cd), device=self.device, dtype=torch.float32)
        off = 0
        for p in parts:
            _lib.check(self._lib.dm_op_copy_channels_nhwc(_lib.ptr(p), p.shape[3], _lib.ptr(y), Cd, off, B * H * W,
                                                          self._stream()))
            off += p.shape[3]
        return y

    # ---- blocks (torchvision/models/inception.py; pytorch_fid/inception.py for the pools of the FID variant) ------
    def _pool_branch(self, x):
        return self._pool(x, 3, 1, 1, POOL_AVG_VALID if self.variant == "fid" else POOL_AVG)

    def _a(self, p, x):
        b1 = self._conv(p + ".branch1x1", x)
        b5 = self._conv(p + ".branch5x5_2", self._conv(p + ".branch5x5_1", x))
        b3 = self._conv(p + ".branch3x3dbl_3", self._conv(p + ".branch3x3dbl_2", self._conv(p + ".branch3x3dbl_1", x)))
        bp = self._conv(p + ".branch_pool", self._pool_branch(x))
        return self._cat([b1, b5, b3, bp])

    def _b(self, p, x):
        b3 = self._conv(p + ".branch3x3", x)
        bd = self._conv(p + ".branch3x3dbl_3", self._conv(p + ".branch3x3dbl_2", self._conv(p + ".branch3x3dbl_1", x)))
        return self._cat([b3, bd, self._pool(x, 3, 2, 0, POOL_MAX)])

    def _c(self, p, x):
        b1 = self._conv(p + ".branch1x1", x)
        b7 = self._conv(p + ".branch7x7_3", self._conv(p + ".branch7x7_2", self._conv(p + ".branch7x7_1", x)))
        bd = x
        for i in range(1, 6):
            bd = self._conv(f"{p}.branch7x7dbl_{i}", bd)
        bp = self._conv(p + ".branch_pool", self._pool_branch(x))
        return self._cat([b1, b7, bd, bp])

    def _d(self, p, x):
        b3 = self._conv(p + ".branch3x3_2", self._conv(p + ".branch3x3_1", x))
        b7 = x
        for i in range(1, 5):
            b7 = self._conv(f"{p}.branch7x7x3_{i}", b7)
        return self._cat([b3, b7, self._pool(x, 3, 2, 0, POOL_MAX)])

    def _e(self, p, x, pool_mode):
        b1 = self._conv(p + ".branch1x1", x)
        b3 = self._conv(p + ".branch3x3_1", x)
        b3a, b3b = self._conv(p + ".branch3x3_2a", b3), self._conv(p + ".branch3x3_2b", b3)
        bd = self._conv(p + ".branch3x3dbl_2", self._conv(p + ".branch3x3dbl_1", x))
        bda, bdb = self._conv(p + ".branch3x3dbl_3a", bd), self._conv(p + ".branch3x3dbl_3b", bd)
        bp = self._conv(p + ".branch_pool", self._pool(x, 3, 1, 1, pool_mode))
        return self._cat([b1, b3a, b3b, bda, bdb, bp])

    def _input(self, x, scale, shift, size):
        x = x.to(self.device, torch.float32).contiguous()
        B, Cc, H, W = x.shape
        assert Cc == 3, "InceptionV3 takes RGB images"
        Ho, Wo = size if size is not None else (H, W)
        y = torch.empty((B, Ho, Wo, 3), device=self.device, dtype=torch.float32)
        sc = torch.tensor(scale, device=self.device, dtype=torch.float32)
        sh = torch.tensor(shift, device=self.device, dtype=torch.float32)
        _lib.check(self._lib.dm_op_resize_bilinear(_lib.ptr(x), _lib.ptr(y), B, 3, H, W, Ho, Wo, _lib.ptr(sc),
                                                   _lib.ptr(sh), self._stream()))
        return y

    def _trunk(self, x, want: Sequence[int]):
        """x NHWC (B, 299, 299, 3) -> {block index: NHWC feature map}."""
        fid = self.variant == "fid"
        outs = {}
        x = self._conv("Conv2d_1a_3x3", x)
        x = self._conv("Conv2d_2a_3x3", x)
        x = self._conv("Conv2d_2b_3x3", x)
        x = self._pool(x, 3, 2, 0, POOL_MAX)
        outs[0] = x
        if max(want) >= 1:
            x = self._conv("Conv2d_3b_1x1", x)
            x = self._conv("Conv2d_4a_3x3", x)
            x = self._pool(x, 3, 2, 0, POOL_MAX)
            outs[1] = x
        if max(want) >= 2:
            for name in ("Mixed_5b", "Mixed_5c", "Mixed_5d"):
                x = self._a(name, x)
            x = self._b("Mixed_6a", x)
            for name in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
                x = self._c(name, x)
            outs[2] = x
        if max(want) >= 3:
            x = self._d("Mixed_7a", x)
            x = self._e("Mixed_7b", x, POOL_AVG_VALID if fid else POOL_AVG)
            x = self._e("Mixed_7c", x, POOL_MAX if fid else POOL_AVG)
            outs[3] = x
        return outs

    def _global_pool(self, x):
        B, H, W, Cc = x.shape
        y = torch.empty((B, Cc), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.dm_op_global_avgpool(_lib.ptr(x), _lib.ptr(y), B, H * W, Cc, self._stream()))
        return y

    @torch.inference_mode()
    def __call__(self, inp: torch.Tensor) -> List[torch.Tensor]:
        """pytorch_fid.InceptionV3.forward: a list with one NCHW feature map per requested block."""
        assert self.variant == "fid", "feature blocks are the pytorch_fid surface; use logits() for torchvision"
        scale, shift = ((2.0,) * 3, (-1.0,) * 3) if self.normalize_input else ((1.0,) * 3, (0.0,) * 3)
        x = self._input(inp, scale, shift, (299, 299) if self.resize_input else None)
        maps = self._trunk(x, self.output_blocks)
        out = []
        for idx in self.output_blocks:
            m = maps[idx]
            if idx == 3:  # the last block of pytorch_fid ends with AdaptiveAvgPool2d((1, 1))
                out.append(self._global_pool(m)[:, :, None, None])
            else:
                out.append(m.permute(0, 3, 1, 2).contiguous())
        return out

    forward = __call__

    @torch.inference_mode()
    def logits(self, x: torch.Tensor) -> torch.Tensor:
        """``inception_v3(weights=..., aux_logits=True).eval()(x)`` for x already normalised with the ImageNet mean /
        std at 299x299 (what inception_score_evaluation.py:80-89 feeds it): transform_input, trunk, pool, fc."""
        assert self.variant == "torchvision"
        # torchvision Inception3._transform_input (transform_input=True whenever pretrained weights are loaded)
        scale = (0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5)
        shift = ((0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5)
        m = self._trunk(self._input(x, scale, shift, None), (3,))[3]
        f = self._global_pool(m)
        W, b = self._fc
        y = torch.empty((f.shape[0], W.shape[0]), device=self.device, dtype=torch.float32)
        _lib.check(self._lib.dm_op_linear(_lib.ptr(f), _lib.ptr(W), _lib.ptr(b), _lib.ptr(y), f.shape[0], W.shape[1],
                                          W.shape[0], self._stream()))
        return y


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6) -> float:
    """``pytorch_fid.fid_score.calculate_frechet_distance`` (host, scipy): |mu1 - mu2|^2 + Tr(S1 + S2 - 2 sqrt(S1 S2))."""
    from scipy import linalg

    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


class FIDEvaluation:
    """fid_evaluation.py:15-133 with the Inception features computed by the HIP path."""

    def __init__(self, batch_size, dl, sampler, channels=3, accelerator=None, stats_dir="./results", device="cuda:0",
                 num_fid_samples=50000, inception_block_idx=2048, inception_state_dict=None, allow_synthetic=False):
        self.batch_size, self.n_samples, self.device, self.channels = batch_size, num_fid_samples, device, channels
        self.dl, self.sampler, self.stats_dir = dl, sampler, stats_dir
        self.print_fn = print if accelerator is None else accelerator.print
        assert inception_block_idx in InceptionV3.BLOCK_INDEX_BY_DIM
        self.inception_v3 = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[inception_block_idx]],
                                        state_dict=inception_state_dict, device=device,
                                        allow_synthetic=allow_synthetic)
        self.dataset_stats_loaded = False

    def calculate_inception_features(self, samples):
        if self.channels == 1:
            samples = samples.expand(-1, 3, -1, -1)
        features = self.inception_v3(samples)[0]
        if features.size(2) != 1 or features.size(3) != 1:
            features = features.mean(dim=(2, 3), keepdim=True)
        return features[:, :, 0, 0]

    # -- statistics of a set of images: what FID compares ------------------------------------------------------------------
    def _moments(self, batches):
        """(mean vector, covariance matrix) of the pooled Inception features of an iterable of image batches."""
        rows = [self.calculate_inception_features(b) for b in batches]
        table = torch.cat(rows, dim=0).cpu().numpy()
        return table.mean(axis=0), np.cov(table, rowvar=False)

    def _real_batches(self):
        """Up to ``ceil(n_samples / batch_size)`` batches of the real-data iterator, on the device; a short loader just ends."""
        for _ in range(math.ceil(self.n_samples / self.batch_size)):
            batch = next(self.dl, None)
            if batch is None:
                return
            yield batch.to(self.device)

    def load_or_precalc_dataset_stats(self):
        """fid_evaluation.py:62-98: the real data's (m2, s2), from ``<stats_dir>/dataset_stats.npz`` (keys ``m2`` / ``s2``,
        the reference's cache format) when that file exists, else computed from the loader and cached there.  With synthetic
        Inception weights the cache is neither read nor written: its numbers would not mean anything to a later run."""
        cache = os.path.join(self.stats_dir, "dataset_stats")
        use_cache = not self.inception_v3.synthetic_weights
        if use_cache and os.path.exists(cache + ".npz"):
            with np.load(cache + ".npz") as stored:
                self.m2, self.s2 = stored["m2"], stored["s2"]
            self.print_fn("Dataset stats loaded from disk.")
        else:
            self.print_fn(f"Stacking Inception features for {self.n_samples} samples from the real dataset.")
            self.m2, self.s2 = self._moments(self._real_batches())
            if use_cache:
                os.makedirs(self.stats_dir, exist_ok=True)
                np.savez_compressed(cache, m2=self.m2, s2=self.s2)
                self.print_fn(f"Dataset stats cached to {cache}.npz for future use.")
        self.dataset_stats_loaded = True

    @torch.inference_mode()
    def fid_score(self, fake_samples):
        """fid_evaluation.py:106-133: Frechet distance between the generated images' feature moments and the real data's."""
        if not self.dataset_stats_loaded:
            self.load_or_precalc_dataset_stats()
        self.sampler.eval()
        self.print_fn(f"Stacking Inception features for {self.n_samples} generated samples.")
        step = self.batch_size
        m1, s1 = self._moments(fake_samples[k:k + step] for k in range(0, len(fake_samples), step))
        return calculate_frechet_distance(m1, s1, self.m2, self.s2)


class InceptionScoreEvaluation:
    """inception_score_evaluation.py:11-113 with the logits computed by the HIP path."""

    def __init__(self, batch_size, sampler, channels=3, accelerator=None, stats_dir="./results", device="cuda:0",
                 num_samples=50000, inception_state_dict=None, allow_synthetic=False):
        self.batch_size, self.n_samples, self.device, self.channels = batch_size, num_samples, device, channels
        self.sampler, self.stats_dir = sampler, stats_dir
        self.print_fn = print if accelerator is None else accelerator.print
        self.inception_model = InceptionV3(variant="torchvision", state_dict=inception_state_dict, device=device,
                                           allow_synthetic=allow_synthetic)
        os.makedirs(stats_dir, exist_ok=True)
        self.log_path = os.path.join(stats_dir, "inception_score_log.txt")

    def _prepare(self, images):
        """What torchvision's classifier expects: three channels in [0, 1] at 299x299, then the ImageNet statistics
        (inception_score_evaluation.py:66-86)."""
        x = images.to(self.device)
        if self.channels == 1:
            x = x.expand(-1, 3, -1, -1)
        if x.min() < 0:  # [-1, 1] -> [0, 1]
            x = (x + 1) / 2.0
        if tuple(x.shape[-2:]) != (299, 299):
            x = torch.nn.functional.interpolate(x, size=(299, 299), mode="bilinear", align_corners=False)
        imagenet = torch.tensor([[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]], device=self.device).view(2, 1, 3, 1, 1)
        return (x - imagenet[0]) / imagenet[1]

    @torch.inference_mode()
    def calculate_inception_score(self, fake_samples):
        """inception_score_evaluation.py:52-104: exp(mean_x KL(p(y|x) || p(y))) over the generated images."""
        self.sampler.eval()
        n = fake_samples.shape[0]
        self.print_fn(f"Calculating Inception Score on {n} generated samples.")
        probs = torch.cat([torch.softmax(self.inception_model.logits(self._prepare(fake_samples[k:k + self.batch_size])), dim=1).cpu()
                           for k in range(0, n, self.batch_size)], dim=0)
        marginal = probs.mean(dim=0)
        tiny = 1e-10
        kl_per_image = (probs * ((probs + tiny).log() - (marginal + tiny).log())).sum(dim=1)
        inception_score = math.exp(kl_per_image.mean().item())
        try:
            with open(self.log_path, "a") as f:
                f.write(f"{inception_score}\n")
        except Exception as e:  # noqa: BLE001 -- the reference only warns here
            self.print_fn("Warning: could not write Inception Score to log file:", e)
        return inception_score
