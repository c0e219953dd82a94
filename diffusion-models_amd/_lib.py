"""ctypes binding of libdm_hip.so (C ABI declared in include/dm_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails the
error is raised, never swallowed.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# DM_LIB selects another build of the same ABI (tools/conv_stamps.py loads the stamped diagnostic build)
LIB_PATH = os.environ.get("DM_LIB") or os.path.join(_HERE, "libdm_hip.so")
DM_MAX_STAGES = 8
DM_COEFS = 8
ABI_VERSION = 5

# every symbol include/dm_hip.h declares (tests check the library exports all of them)
EXPORTS = (
    "dm_last_error", "dm_abi_version",
    "dm_unet_create", "dm_unet_destroy", "dm_unet_set_param", "dm_unet_missing_params", "dm_unet_get_param_host", "dm_unet_finalize",
    "dm_unet_update_param", "dm_unet_refresh", "dm_unet_graph_captures", "dm_unet_workspace_bytes",
    "dm_unet_forward", "dm_sample", "dm_sample_cond", "dm_sample_ex", "dm_randn",
    "dm_decoder_create", "dm_decoder_destroy", "dm_decoder_set_param", "dm_decoder_missing_params",
    "dm_decoder_finalize", "dm_decoder_forward",
    "dm_encoder_create", "dm_encoder_destroy", "dm_encoder_set_param", "dm_encoder_missing_params",
    "dm_encoder_finalize", "dm_encoder_forward",
    "dm_op_conv2d", "dm_op_downsample", "dm_op_rmsnorm", "dm_op_block", "dm_op_linear_attention",
    "dm_op_attention", "dm_op_sampler_update",
    "dm_conv_create", "dm_conv_destroy", "dm_conv_forward", "dm_op_pool2d", "dm_op_resize_bilinear",
    "dm_op_copy_channels_nhwc", "dm_op_global_avgpool", "dm_op_linear",
    "dm_unet_train_enable", "dm_unet_grad_floats", "dm_unet_grads_flat", "dm_unet_train_buckets", "dm_unet_train_bucket", "dm_unet_get_grad", "dm_unet_loss_backward", "dm_unet_loss_backward_ex", "dm_op_q_sample", "dm_op_linear_bwd",
    "dm_op_offset_noise", "dm_op_cdist", "dm_op_gather_rows", "dm_op_lincomb", "dm_op_mask_mix",
    "dm_unet_optimizer_step", "dm_unet_train_scalar", "dm_unet_ema_update", "dm_unet_get_param", "dm_unet_set_train_tensor", "dm_unet_adam_step",
    "dm_unet_train_sync", "dm_unet_check_device_pack",
    "dm_unet_train_dropout", "dm_op_dropout_mask",
    "dm_op_conv2d_bwd", "dm_op_downsample_bwd", "dm_op_block_bwd", "dm_op_rmsnorm_bwd", "dm_op_linear_attention_bwd",
    "dm_op_attention_bwd",
    "dm_profile_enable", "dm_profile_read",
)


class UnetCfg(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("init_dim", C.c_int32), ("out_dim", C.c_int32), ("channels", C.c_int32),
        ("input_channels", C.c_int32), ("n_stages", C.c_int32),
        ("dim_mults", C.c_int32 * DM_MAX_STAGES), ("full_attn", C.c_int32 * DM_MAX_STAGES),
        ("attn_heads", C.c_int32), ("attn_dim_head", C.c_int32),
        ("text_mode", C.c_int32), ("text_emb_dim", C.c_int32), ("sinusoidal_theta", C.c_float),
        ("learned_sinusoidal_dim", C.c_int32), ("attn_heads_stage", C.c_int32 * DM_MAX_STAGES),
    ]


class DecoderCfg(C.Structure):
    _fields_ = [
        ("ch", C.c_int32), ("out_ch", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * DM_MAX_STAGES),
        ("num_res_blocks", C.c_int32), ("n_attn_res", C.c_int32), ("attn_resolutions", C.c_int32 * DM_MAX_STAGES),
        ("resolution", C.c_int32), ("z_channels", C.c_int32), ("embed_dim", C.c_int32),
    ]


class EncoderCfg(C.Structure):
    _fields_ = [
        ("ch", C.c_int32), ("in_channels", C.c_int32), ("n_levels", C.c_int32), ("ch_mult", C.c_int32 * DM_MAX_STAGES),
        ("num_res_blocks", C.c_int32), ("n_attn_res", C.c_int32), ("attn_resolutions", C.c_int32 * DM_MAX_STAGES),
        ("resolution", C.c_int32), ("z_channels", C.c_int32), ("embed_dim", C.c_int32), ("n_embed", C.c_int32),
        ("double_z", C.c_int32),
    ]


class SampleArgs(C.Structure):
    """dm_sample_args (include/dm_hip.h)."""
    _fields_ = [
        ("kind", C.c_int32), ("objective", C.c_int32), ("self_condition", C.c_int32), ("n_steps", C.c_int32),
        ("times_host", C.POINTER(C.c_int64)), ("coefs_host", C.POINTER(C.c_float)),
        ("x_T", C.c_void_p), ("noise", C.c_void_p), ("seed", C.c_uint64), ("sample_offset", C.c_uint64),
        ("ctx", C.c_void_p), ("ctx_tokens", C.c_int32), ("cond_channels", C.c_int32), ("cond", C.c_void_p),
        ("out", C.c_void_p), ("all_steps", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("unnormalize", C.c_int32), ("use_graph", C.c_int32),
        ("reserved_", C.c_int32), ("stream", C.c_void_p),
    ]


OBJECTIVES = {"pred_noise": 0, "pred_x0": 1, "pred_v": 2}


class ProfileRow(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("total_flops", C.c_double), ("total_bytes", C.c_double)]


_lib: Optional[C.CDLL] = None


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64, u64, fp = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_void_p
    lib.dm_last_error.restype = C.c_char_p
    lib.dm_last_error.argtypes = []
    lib.dm_abi_version.restype = i32
    lib.dm_unet_create.argtypes = [C.POINTER(UnetCfg), i32, C.POINTER(vp)]
    lib.dm_unet_destroy.argtypes = [vp]
    lib.dm_unet_destroy.restype = None
    lib.dm_unet_set_param.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    lib.dm_unet_missing_params.argtypes = [vp]
    lib.dm_unet_finalize.argtypes = [vp]
    lib.dm_unet_get_param_host.argtypes = [vp, C.c_char_p, vp, i64]
    lib.dm_unet_forward.argtypes = [vp, fp, vp, fp, i32, fp, i32, i32, i32, vp]
    lib.dm_unet_update_param.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    lib.dm_unet_refresh.argtypes = [vp]
    lib.dm_unet_graph_captures.argtypes = [vp]
    lib.dm_unet_workspace_bytes.argtypes = [vp]
    lib.dm_unet_workspace_bytes.restype = i64
    lib.dm_sample.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(C.c_float), fp, fp, u64, u64, fp, i32, fp, fp,
                              i32, i32, i32, i32, i32, vp]
    lib.dm_sample_cond.argtypes = [vp, i32, i32, C.POINTER(i64), C.POINTER(C.c_float), fp, fp, u64, u64, fp, i32, fp,
                                   i32, fp, fp, i32, i32, i32, i32, i32, vp]
    lib.dm_sample_ex.argtypes = [vp, C.POINTER(SampleArgs)]
    lib.dm_randn.argtypes = [fp, i64, u64, u64, u64, vp]
    lib.dm_decoder_create.argtypes = [C.POINTER(DecoderCfg), i32, C.POINTER(vp)]
    lib.dm_decoder_destroy.argtypes = [vp]
    lib.dm_decoder_destroy.restype = None
    lib.dm_decoder_set_param.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    lib.dm_decoder_missing_params.argtypes = [vp]
    lib.dm_decoder_finalize.argtypes = [vp]
    lib.dm_decoder_forward.argtypes = [vp, fp, fp, i32, i32, i32, vp]
    lib.dm_encoder_create.argtypes = [C.POINTER(EncoderCfg), i32, C.POINTER(vp)]
    lib.dm_encoder_destroy.argtypes = [vp]
    lib.dm_encoder_destroy.restype = None
    lib.dm_encoder_set_param.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    lib.dm_encoder_missing_params.argtypes = [vp]
    lib.dm_encoder_finalize.argtypes = [vp]
    lib.dm_encoder_forward.argtypes = [vp, fp, fp, fp, vp, i32, i32, i32, vp]
    lib.dm_op_conv2d.argtypes = [fp, i32, fp, i32, fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.dm_op_downsample.argtypes = [fp, i32, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.dm_op_rmsnorm.argtypes = [fp, fp, fp, i32, i32, i32, i32, vp]
    lib.dm_op_block.argtypes = [fp, i32, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.dm_op_linear_attention.argtypes = [fp, fp, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, vp]
    lib.dm_op_attention.argtypes = [fp, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, vp]
    lib.dm_op_sampler_update.argtypes = [i32, i32, fp, fp, fp, C.POINTER(C.c_float), fp, fp, i64, vp]
    lib.dm_conv_create.argtypes = [fp, fp, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.dm_conv_destroy.argtypes = [vp]
    lib.dm_conv_destroy.restype = None
    lib.dm_conv_forward.argtypes = [vp, fp, i32, i32, i32, i32, fp, vp]
    lib.dm_op_pool2d.argtypes = [fp, fp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.dm_op_resize_bilinear.argtypes = [fp, fp, i32, i32, i32, i32, i32, i32, fp, fp, vp]
    lib.dm_op_copy_channels_nhwc.argtypes = [fp, i32, fp, i32, i32, i64, vp]
    lib.dm_op_global_avgpool.argtypes = [fp, fp, i32, i32, i32, vp]
    lib.dm_op_linear.argtypes = [fp, fp, fp, fp, i32, i32, i32, vp]
    lib.dm_unet_train_enable.argtypes = [vp]
    lib.dm_unet_grad_floats.argtypes = [vp]
    lib.dm_unet_grad_floats.restype = i64
    lib.dm_unet_get_grad.argtypes = [vp, C.c_char_p, fp, vp]
    lib.dm_unet_grads_flat.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    lib.dm_unet_train_buckets.argtypes = [vp, i32]
    lib.dm_unet_train_bucket.argtypes = [vp, i32, C.POINTER(i64), C.POINTER(i64), i32, vp]
    lib.dm_unet_loss_backward.argtypes = [vp, fp, C.POINTER(i64), C.POINTER(C.c_float), fp, fp, fp, i32, fp, i32, i32, i32,
                                          C.c_float, i32, C.POINTER(C.c_float), fp, i32, i32, i32, vp]
    lib.dm_unet_loss_backward_ex.argtypes = [vp, C.POINTER(TrainArgs)]
    lib.dm_unet_optimizer_step.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), vp]
    lib.dm_unet_train_scalar.argtypes = [vp, i32, fp, vp]
    lib.dm_unet_ema_update.argtypes = [vp, C.c_float, i32, vp]
    lib.dm_unet_get_param.argtypes = [vp, C.c_char_p, i32, fp, vp]
    lib.dm_unet_set_train_tensor.argtypes = [vp, C.c_char_p, i32, fp, vp]
    lib.dm_unet_adam_step.argtypes = [vp, C.c_longlong]
    lib.dm_unet_adam_step.restype = C.c_longlong
    lib.dm_unet_train_sync.argtypes = [vp]
    lib.dm_unet_train_dropout.argtypes = [vp, C.c_float, u64]
    lib.dm_op_dropout_mask.argtypes = [fp, i64, C.c_float, u64, u64, i32, vp]
    lib.dm_unet_check_device_pack.argtypes = [vp]
    lib.dm_op_q_sample.argtypes = [fp, fp, C.POINTER(C.c_float), fp, i32, i32, vp]
    lib.dm_op_offset_noise.argtypes = [fp, fp, C.c_float, i32, i32, vp]
    lib.dm_op_lincomb.argtypes = [fp, fp, C.POINTER(C.c_float), fp, i32, i64, i32, i32, vp]
    lib.dm_op_mask_mix.argtypes = [fp, fp, fp, fp, i64, vp]
    lib.dm_op_cdist.argtypes = [fp, fp, fp, i32, i32, i64, vp]
    lib.dm_op_gather_rows.argtypes = [fp, C.POINTER(i64), fp, i32, i64, vp]
    lib.dm_op_linear_bwd.argtypes = [fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.dm_op_conv2d_bwd.argtypes = [fp, i32, fp, i32, fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.dm_op_downsample_bwd.argtypes = [fp, i32, fp, fp, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.dm_op_block_bwd.argtypes = [fp, i32] + [fp] * 12 + [i32, i32, i32, i32, vp]
    lib.dm_op_rmsnorm_bwd.argtypes = [fp, fp, fp, fp, fp, i32, i32, i32, i32, vp]
    lib.dm_op_linear_attention_bwd.argtypes = [fp] * 15 + [i32] * 6 + [vp]
    lib.dm_op_attention_bwd.argtypes = [fp] * 13 + [i32] * 6 + [vp]
    lib.dm_profile_enable.argtypes = [i32]
    lib.dm_profile_read.argtypes = [C.POINTER(ProfileRow), i32, C.POINTER(i32)]


class TrainArgs(C.Structure):
    """``dm_train_args`` of include/dm_hip.h (dm_unet_loss_backward_ex)."""
    _fields_ = [("x_start", C.c_void_p), ("t_host", C.POINTER(C.c_int64)), ("coef_host", C.POINTER(C.c_float)),
                ("coef_stride", C.c_int), ("noise", C.c_void_p), ("noise_q", C.c_void_p), ("cond", C.c_void_p),
                ("cond_channels", C.c_int), ("ctx", C.c_void_p), ("ctx_tokens", C.c_int), ("self_cond", C.c_int),
                ("objective", C.c_int), ("loss_scale", C.c_float), ("accumulate", C.c_int),
                ("loss_out_host", C.POINTER(C.c_float)), ("model_out", C.c_void_p), ("B", C.c_int), ("H", C.c_int),
                ("W", C.c_int), ("stream", C.c_void_p), ("loss_terms", C.c_int), ("kl_scale", C.c_float)]


def load() -> C.CDLL:
    """Load libdm_hip.so; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C diffusion-models_amd/csrc`). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    _declare(lib)
    if lib.dm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libdm_hip.so ABI {lib.dm_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def profile_enable(on: bool) -> None:
    check(load().dm_profile_enable(1 if on else 0))


def profile_read():
    """[{kernel, launches, total_ms, total_flops, total_bytes}] since the last read (synchronises)."""
    rows = (ProfileRow * 64)()
    n = C.c_int(0)
    check(load().dm_profile_read(rows, 64, C.byref(n)))
    return [dict(kernel=rows[i].kernel.decode(), launches=int(rows[i].launches), total_ms=float(rows[i].total_ms),
                 total_flops=float(rows[i].total_flops), total_bytes=float(rows[i].total_bytes))
            for i in range(n.value)]


def check(rc: int) -> None:
    if rc != 0:
        msg = load().dm_last_error().decode(errors="replace")
        raise RuntimeError(f"libdm_hip: {msg}")


def ptr(t) -> Optional[int]:
    """Device (or host) address of a contiguous fp32 / int64 torch tensor, None passes NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous at the C ABI"
    return t.data_ptr()
