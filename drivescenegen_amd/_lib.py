"""ctypes binding of libdsg.so (C ABI: include/dsg.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this module
raises, and every op raises when handed a non-GPU tensor.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (imported first so libdsg.so binds to the HIP runtime torch already loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSG_LIB_PATH: an alternative build of the same ABI (kernel A/B comparisons in tools/); default is the in-tree build
LIB_PATH = os.environ.get("DSG_LIB_PATH") or os.path.join(_HERE, "lib", "libdsg.so")

OK = 0
ERR_NAMES = {-1: "DSG_ERR_INVALID_ARG", -2: "DSG_ERR_UNSUPPORTED_SHAPE", -3: "DSG_ERR_WORKSPACE_TOO_SMALL",
             -4: "DSG_ERR_HIP", -5: "DSG_ERR_NOT_READY"}


class DsgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class ConvArgs(C.Structure):
    """Mirror of ``dsg_conv_args`` (include/dsg.h)."""
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("n", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("upsample", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("cout", C.c_int32),
        ("weight", C.c_void_p), ("weight_cout_stride", C.c_int32),
        ("bias", C.c_void_p), ("gn_scale_shift", C.c_void_p),
        ("silu", C.c_int32),
        ("temb", C.c_void_p), ("temb_stride", C.c_int32),
        ("residual", C.c_void_p), ("dst", C.c_void_p), ("pool2", C.c_int32),
        ("weight_h2", C.c_void_p),
        ("weight_h2_cout_stride", C.c_int32),
        ("weight_h2_fold", C.c_void_p),
        ("stats_out", C.c_void_p),
        ("src_layout", C.c_int32), ("dst_layout", C.c_int32),
        ("weight_h2_s2", C.c_void_p),
        ("compute_dtype", C.c_int32),
        ("src_bound", C.c_void_p), ("src_bound1", C.c_void_p),
        ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_size_t),
        ("sc_src0", C.c_void_p), ("sc_src1", C.c_void_p), ("sc_c0", C.c_int32), ("sc_c1", C.c_int32),
        ("sc_weight_h2", C.c_void_p), ("sc_bias", C.c_void_p), ("sc_src_bound", C.c_void_p), ("sc_src_bound1", C.c_void_p),
        ("src_operand", C.c_void_p),
        ("gnb_x0", C.c_void_p), ("gnb_x1", C.c_void_p), ("gnb_c0", C.c_int32), ("gnb_ss", C.c_void_p), ("gnb_silu", C.c_int32),
        ("s2_window4", C.c_int32),
    ]


class PackJob(C.Structure):
    """Mirror of ``dsg_pack_job`` (include/dsg.h)."""
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("ksize", C.c_int32),
                ("kind", C.c_int32), ("dtype", C.c_int32), ("n_total", C.c_int32), ("n_off", C.c_int32), ("n_pad", C.c_int32)]


class ConvWgradArgs(C.Structure):
    """Mirror of ``dsg_conv_wgrad_args`` (include/dsg.h)."""
    _fields_ = [
        ("src0", C.c_void_p), ("src1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("n", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("upsample", C.c_int32), ("ksize", C.c_int32), ("stride", C.c_int32), ("cout", C.c_int32),
        ("dy", C.c_void_p), ("dy_ctotal", C.c_int32), ("dy_coff", C.c_int32),
        ("gn_scale_shift", C.c_void_p), ("silu", C.c_int32),
        ("dw", C.c_void_p), ("force_direct", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("compute_dtype", C.c_int32),
        ("dy_sums", C.c_void_p), ("dy_sums_stride", C.c_int32),
        ("dy_bias_grad", C.c_void_p),
    ]


class UNetConfig(C.Structure):
    """Mirror of ``dsg_unet_config`` (include/dsg.h)."""
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32),
        ("sample_h", C.c_int32), ("sample_w", C.c_int32),
        ("layers_per_block", C.c_int32), ("num_blocks", C.c_int32),
        ("block_out_channels", C.c_int32 * 8), ("down_attn", C.c_int32 * 8), ("up_attn", C.c_int32 * 8),
        ("norm_num_groups", C.c_int32), ("norm_eps", C.c_float),
        ("attention_head_dim", C.c_int32), ("add_attention", C.c_int32),
        ("compute_dtype", C.c_int32),
        ("flags", C.c_uint32),
    ]


UNET_BATCH_INVARIANT = 1  # dsg_unet_config.flags

_vp, _i32, _i64, _f32, _sz, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_double

# dsg_dtype (include/dsg.h): arithmetic type of the matrix-core products / storage type of channel-blocked tensors
DSG_F32, DSG_BF16, DSG_F16 = 0, 1, 2
DTYPE_CODES = {"fp32": DSG_F32, "bf16": DSG_BF16, "fp16": DSG_F16}
TORCH_DTYPES = {DSG_F32: torch.float32, DSG_BF16: torch.bfloat16, DSG_F16: torch.float16}

# name -> argtypes; every function returns int32 status unless noted.  This table is the single
# Python-side statement of the ABI; tests/test_abi.py checks it against include/dsg.h.
SIGNATURES = {
    "dsg_conv2d_stats_tiles": [_vp, _vp],
    "dsg_conv2d_splitk_bytes": [_vp, C.POINTER(_sz)],
    "dsg_conv_weight_pack": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_conv_weight_pack_bytes": [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_sz)],
    "dsg_conv_weight_pack_batch_items": [C.POINTER(PackJob), C.POINTER(_i64)],
    "dsg_conv_weight_pack_batch": [_vp, _vp, _i32, _i64, _vp],
    "dsg_layout_convert_dt": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_gn_channel_stats_blocked_dt": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp],
    "dsg_unscale_check": [_vp, _i64, _f32, _vp, _vp],
    "dsg_range_bound_from_stats": [_vp, _i32, _i32, _i32, _vp, _vp],
    "dsg_gn_finalize_parts_bound": [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "dsg_abs_max": [_vp, _i64, _vp, _vp],
    "dsg_gn_bwd_blocked": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                           _vp, _vp, _i32, _vp],
    "dsg_gn_bwd_blocked_add2": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                _vp, _vp, _vp, _i32, _vp],
    "dsg_gn_bwd_blocked_splits": [_i32],
    "dsg_gn_bwd_add2": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                        _vp, _vp, _vp, _i32, _vp],
    "dsg_gn_bwd_parts": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                         _vp, _vp, _vp, _i32, _vp],
    "dsg_gn_bwd_blocked_parts": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _vp, _vp, _vp, _i32, _vp, _i32, _vp],
    "dsg_channel_sums_blocked": [_vp, _i32, _i32, _i32, _vp, _i32, _i32, _vp],
    "dsg_add_dt": [_vp, _vp, _i64, _vp, _i32, _vp],
    "dsg_upsample_nearest2x_blocked": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "dsg_sumpool2x2_blocked": [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "dsg_conv_weight_relayout_h2_fold": [_vp, _vp, _i32, _i32, _vp],
    "dsg_conv_weight_relayout_h2_s2": [_vp, _vp, _i32, _i32, _vp],
    "dsg_upsample_nearest2x": [_vp, _vp, _i64, _i32, _i32, _vp],
    "dsg_sumpool2x2": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "dsg_rasterize_boxes": [_vp, _i32, _vp, _i32, _i32, _f32, _f32, _f32, _vp],
    "dsg_gn_finalize_parts": [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp],
    "dsg_gn_finalize_parts_train": [_vp, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "dsg_conv2d_fwd": [C.POINTER(ConvArgs), _vp],
    "dsg_conv2d_fuses_shortcut": [C.POINTER(ConvArgs), C.POINTER(_i32)],
    "dsg_conv2d_takes_operand": [C.POINTER(ConvArgs), C.POINTER(_i32)],
    "dsg_conv2d_gnb_supported": [C.POINTER(ConvArgs), C.POINTER(_i32)],
    "dsg_conv_operand_bytes": [_i32, _i32, _i32, _i32, _i32, C.POINTER(_sz)],
    "dsg_conv_operand_prepare": [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
    "dsg_layout_convert": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "dsg_conv2d_fwd_direct": [C.POINTER(ConvArgs), _vp],
    "dsg_conv_weight_relayout": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_conv_weight_relayout_dgrad": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_conv_weight_relayout_h2": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_conv_weight_relayout_h2_dgrad": [_vp, _vp, _i32, _i32, _i32, _vp],
    "dsg_gn_channel_stats": [_vp, _i32, _vp, _i32, _i32, _i32, _vp, _vp],
    "dsg_gn_channel_stats_blocked": [_vp, _i32, _i32, _i32, _i32, _vp, _vp],
    "dsg_gn_finalize": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp],
    "dsg_gn_apply": [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp],
    "dsg_attention_fwd": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "dsg_attention_fwd_dt": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_attention_fwd_blocked": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_time_embed_fwd": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "dsg_linear_fwd": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp],
    "dsg_add_noise": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp],
    "dsg_host_device_pointer": [_vp, C.POINTER(_vp)],
    "dsg_philox_u32": [_vp, _i64, C.c_uint64, C.c_uint64, _vp],
    "dsg_philox_normal": [_vp, _i64, C.c_uint64, C.c_uint64, _vp],
    "dsg_add_noise_philox": [_vp, _vp, _vp, _vp, _vp, _i32, _i64, C.c_uint64, C.c_uint64, _vp],
    "dsg_ddpm_step": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _vp],
    "dsg_ddim_step": [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _vp],
    "dsg_postprocess": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "dsg_unet_create": [C.POINTER(UNetConfig), C.POINTER(_vp)],
    "dsg_unet_set_param": [_vp, C.c_char_p, _vp, _i64, _vp],
    "dsg_unet_commit_params": [_vp],
    "dsg_unet_num_params": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "dsg_unet_param_name": [_vp, _i64, C.POINTER(C.c_char_p), C.POINTER(_i64)],
    "dsg_unet_workspace_bytes": [_vp, _i32, C.POINTER(_sz)],
    "dsg_unet_forward": [_vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp],
    "dsg_conv2d_wgrad": [C.POINTER(ConvWgradArgs), _vp],
    "dsg_conv2d_wgrad_workspace_bytes": [C.POINTER(ConvWgradArgs), C.POINTER(_sz)],
    "dsg_gn_finalize_train": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "dsg_gn_bwd": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                   _vp, _vp, _vp],
    "dsg_channel_sums": [_vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "dsg_add": [_vp, _vp, _i64, _vp, _vp],
    "dsg_time_embed_fwd_train": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "dsg_reduce_rows_add": [_vp, _i32, _i32, _i32, _vp, _vp],
    "dsg_attention_fwd_train": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "dsg_attention_fwd_train_dt": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "dsg_attention_bwd_dt": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "dsg_linear_bwd": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "dsg_scale": [_vp, _i64, _vp, _f32, _vp, _vp],
    "dsg_silu_fwd": [_vp, _i64, _vp, _vp],
    "dsg_silu_bwd": [_vp, _vp, _i64, _vp, _vp],
    "dsg_mse_loss": [_vp, _vp, _i64, _f32, _vp, _vp, _vp, _sz, _vp],
    "dsg_l2_norm": [_vp, _i64, _vp, _vp, _sz, _vp],
    "dsg_clip_scale": [_vp, _i64, _vp, _f32, _vp],
    "dsg_adamw_step": [_vp, _vp, _vp, _vp, _i64, _f64, _f64, _f64, _f64, _f64, _i64, _vp, _f32, _vp],
    "dsg_resize_normalize_u8": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _f32, _vp],
    "dsg_resize_normalize_f32": [_vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _f32, _f32, _vp],
    "dsg_png_probe": [C.c_char_p, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)],
    "dsg_png_decode_batch": [C.POINTER(C.c_char_p), _i32, _vp, _i32, _i32, _i32, _i32, C.POINTER(_i32)],
    "dsg_hist_u8": [_vp, _i32, _i32, _i32, _vp, _vp],
    "dsg_mask_lut_u8": [_vp, _i32, _i32, _i32, _i32, _i32, _vp, C.c_uint8, C.c_uint8, _vp, _vp],
    "dsg_prof_enable": [_i32],
    "dsg_set_tuning": [_i32, _i32],
    "dsg_tuning_epoch": [],
    "dsg_prof_dump": [C.c_char_p],
    "dsg_prof_summary": [_i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)],
}
OTHER_SYMBOLS = ["dsg_version", "dsg_last_error", "dsg_unet_destroy"]

_lib = None
_lock = threading.Lock()


def load():
    """Load libdsg.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libdsg.so not found at {LIB_PATH}: build it with `python drivescenegen_amd/csrc/build.py` "
                "(or __graft_entry__.build()). The engine has no CPU fallback.")
        if os.environ.get("DSG_TUNING"):   # (the switches are a test hook the library only honours under DSG_TESTING=1)
            os.environ["DSG_TESTING"] = "1"
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int32
        lib.dsg_version.restype = C.c_int32
        # DSG_TUNING="key=value,...": kernel-selection knobs (dsg_set_tuning) for A/B runs of tests and tools
        for kv in filter(None, os.environ.get("DSG_TUNING", "").split(",")):
            k, v = kv.split("=")
            if lib.dsg_set_tuning(int(k), int(v)) != 0:
                raise RuntimeError(f"DSG_TUNING: bad entry {kv!r}")
        lib.dsg_version.argtypes = []
        lib.dsg_last_error.restype = C.c_char_p
        lib.dsg_last_error.argtypes = []
        lib.dsg_unet_destroy.restype = None
        lib.dsg_unet_destroy.argtypes = [_vp]
        _lib = lib
    return _lib


def check(rc: int):
    if rc != OK:
        raise DsgError(rc, load().dsg_last_error().decode("utf-8", "replace"))


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    """Device pointer of a contiguous fp32/int64 GPU tensor (or None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("drivescenegen_amd: the HIP engine needs GPU tensors (got a CPU tensor); "
                           "there is no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("drivescenegen_amd: tensor must be contiguous")
    return t.data_ptr()
