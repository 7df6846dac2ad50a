"""ctypes binding of libddpm_hip.so (C ABI declared in include/ddpm_hip.h).

There is no CPU fallback: loading fails loudly when the library has not been built, and every call
raises RuntimeError with the library's message when an entry point reports an error.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DDPM_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libddpm_hip.so")     # (the variable: diagnostic builds)

ACT = {"none": 0, None: 0, "gelu": 1, "silu": 2, "relu": 3, "tanh": 4, "sigmoid": 5, "hardsigmoid": 6, "relu6": 7, "leaky_relu": 8, "selu": 9,
       "swish": 10, "hardswish": 11}
ACT_WITH_SCALAR = (10, 11)        # swish, hardswish: one learnable scalar per activation module
CONV_3X3_S1, CONV_3X3_S2, CONV_1X1, CONVT_2X2_S2, DGRAD_3X3_S2, DGRAD_CONVT_2X2 = range(6)
OUT_BF16_NHWC, OUT_F32_NCHW = 0, 1
CONV_KERNEL_NAMES = {0: "conv_mfma_kernel", 1: "conv_mfma_kernel", 2: "conv_small_kernel", 3: "conv_thin_kernel", 4: "conv_img_kernel",
                     5: "conv_pipe_kernel", 6: "pw_stream_kernel"}

_vp, _i, _f, _i64, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_uint32


class ConvDesc(C.Structure):
    _fields_ = [
        ("kind", _i), ("N", _i), ("IH", _i), ("IW", _i),
        ("in0", _vp), ("in1", _vp), ("C0", _i), ("C1", _i),
        ("in_add", _vp), ("in_mask", _vp), ("in_mask_scale", _f),
        ("w", _vp), ("bias", _vp),
        ("out", _vp), ("res", _vp), ("out_mask", _vp), ("out_mask_scale", _f),
        ("Cout", _i), ("CoutW", _i), ("out_cs", _i), ("w_rows", _i),
        ("act", _i), ("out_mode", _i), ("sum_out", _vp),
        ("gn_gamma", _vp), ("gn_beta", _vp), ("gn_out", _vp), ("gn_stats", _vp), ("gn_res", _vp),
        ("gn_groups", _i), ("gn_act", _i), ("gn_eps", _f), ("out2", _vp), ("act2", _i),
        ("gnb_raw", _vp), ("gnb_gamma", _vp), ("gnb_beta", _vp), ("gnb_stats", _vp), ("gnb_part", _vp), ("gnb_draw", _vp),
        ("gnb_groups", _i), ("gnb_act", _i), ("C0_used", _i), ("gn_part", _vp), ("gn_part_groups", _i), ("in_affine", _vp), ("in_affine_act", _i),
        ("gn_act_beta", _vp),
    ]


class GnDesc(C.Structure):
    _fields_ = [
        ("x", _vp), ("gamma", _vp), ("beta", _vp), ("res", _vp), ("mask", _vp), ("mask_scale", _f),
        ("y", _vp), ("stats", _vp), ("dy", _vp), ("part", _vp),
        ("N", _i), ("HW", _i), ("C", _i), ("G", _i), ("act", _i), ("eps", _f), ("scratch", _vp), ("scratch_floats", _i64),
        ("act_beta", _vp), ("dbeta_part", _vp), ("fixed_stats", _i), ("stat_part", _vp), ("stat_parts", _i),
    ]


class SirDesc(C.Structure):
    _fields_ = [(k, _vp) for k in ("x", "y", "we", "be", "g1", "b1", "wd", "bd", "g2", "b2", "wp", "bp", "g3", "b3", "m_in", "m_e", "m_d")] + [
        ("mask_scale", _f), ("eps", _f)] + [(k, _vp) for k in ("raw_e", "y_e", "st1", "raw_d", "y_d", "st2", "raw_o", "st3")] + [
        (k, C.c_int32) for k in ("N", "C", "H", "W", "E", "groups", "act", "pad")] + [("act_b1", _vp), ("act_b2", _vp)]


class SirBwdDesc(C.Structure):
    _fields_ = [(k, _vp) for k in ("dy", "raw_e", "raw_d", "raw_o", "y_e", "st1", "st2", "st3", "g1", "b1", "g2", "b2", "g3", "wpd", "wed",
                                   "wdf", "m_in", "m_e", "m_d", "dx", "d_o", "d_e", "part1", "part2", "part3", "dwpart")] + [
        ("mask_scale", _f)] + [(k, C.c_int32) for k in ("N", "C", "H", "W", "E", "groups", "act")] + [
        (k, _vp) for k in ("act_b1", "act_b2", "dbeta1", "dbeta2")]


EMBED_MAX = 8
_vp8, _i8 = _vp * EMBED_MAX, C.c_int32 * EMBED_MAX


class EmbedDesc(C.Structure):
    _fields_ = [("sin", _vp), ("N", C.c_int32), ("t_dim", C.c_int32), ("act", C.c_int32), ("nmlp", C.c_int32), ("C", _i8),
                ("w0", _vp8), ("b0", _vp8), ("w1", _vp8), ("b1", _vp8), ("out", _vp8), ("raw0", _vp8), ("h", _vp8)]


class EmbedBwdDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("act", C.c_int32), ("nmlp", C.c_int32), ("pad", C.c_int32), ("C", _i8),
                ("dtemb", _vp8), ("raw0", _vp8), ("w1d", _vp8), ("dt16", _vp8), ("dz", _vp8), ("db1", _vp8), ("db0", _vp8)]


class AttnDesc(C.Structure):
    _fields_ = [("q", _vp), ("k", _vp), ("v", _vp), ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("o", _vp), ("ldo", C.c_int32),
                ("lse", _vp), ("dout", _vp), ("ldd", C.c_int32), ("dq", _vp), ("dk", _vp), ("dv", _vp), ("lddq", C.c_int32),
                ("lddk", C.c_int32), ("lddv", C.c_int32), ("N", C.c_int32), ("S", C.c_int32), ("Sk", C.c_int32), ("heads", C.c_int32),
                ("dh", C.c_int32), ("drop_p", _f), ("seed", _vp), ("stream_id", _u32)]


class SampleNormDesc(C.Structure):
    _fields_ = [("a", _vp), ("b", _vp), ("gamma", _vp), ("beta", _vp), ("y", _vp), ("stats", _vp), ("dy", _vp), ("du", _vp), ("da", _vp),
                ("part", _vp), ("N", C.c_int32), ("S", C.c_int32), ("C", C.c_int32), ("act", C.c_int32), ("eps", _f),
                ("act_beta", _vp), ("dbeta_part", _vp)]


class ReduceEntry(C.Structure):
    _fields_ = [("src_off", _i64), ("dst_off", _i64), ("rows", C.c_int32), ("row_stride", C.c_int32), ("width", C.c_int32),
                ("scalar", C.c_int32)]


class RemapEntry(C.Structure):
    _fields_ = [("a", _vp), ("b", _vp), ("S", C.c_int32), ("R", C.c_int32), ("K", C.c_int32), ("Rv", C.c_int32), ("Kv", C.c_int32),
                ("a_r", C.c_int32), ("a_k", C.c_int32), ("off_s", C.c_int32 * 9), ("a_bf16", C.c_int32), ("pad", C.c_int32)]


class WgradDesc(C.Structure):
    _fields_ = [
        ("kind", _i), ("N", _i), ("IH", _i), ("IW", _i),
        ("in0", _vp), ("in1", _vp), ("C0", _i), ("C1", _i),
        ("in_add", _vp), ("in_mask", _vp), ("in_mask_scale", _f),
        ("dy", _vp), ("Cout", _i), ("dy_cs", _i), ("dw", _vp), ("part", _vp), ("tiles_per_split", _i),
    ]


class LdsOp(C.Structure):
    _fields_ = [("type", _i), ("kind", _i), ("src_off", _i), ("src_stride", _i), ("dst_off", _i), ("dst_stride", _i), ("aux_off", _i),
                ("aux_stride", _i), ("w_in", _i), ("w_out", _i), ("cin", _i), ("cout", _i), ("col0", _i), ("groups", _i), ("act", _i),
                ("gstride", _i), ("gcol0", _i), ("pad0", _i), ("eps", _f), ("mscale", _f), ("p", _vp * 6)]


LOP_END, LOP_LOAD, LOP_STORE, LOP_CONV, LOP_GN, LOP_DW, LOP_ADDT, LOP_MASK = range(8)

_PROTOS = {
    "ddpm_set_deterministic": ([_i], _i),
    "ddpm_deterministic": ([], _i),
    "ddpm_unet_lds_arena0": ([], _i),
    "ddpm_unet_lds_reduce_bytes": ([], _i),
    "ddpm_unet_lds_max_bytes": ([], _i),
    "ddpm_unet_lds_check": ([_vp, _i, _i], C.c_longlong),
    "ddpm_unet_lds_run": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_abi_version": ([], _i),
    "ddpm_q_sample": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _u32, _vp], _i),
    "ddpm_reverse_step": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _u32, _i, C.c_uint64, _vp], _i),
    "ddpm_train_head": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp], _i),
    "ddpm_fill_normal": ([_vp, _i64, _vp, _u32, C.c_uint64, _vp], _i),
    "ddpm_elbo_fwd": ([_vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp], _i),
    "ddpm_elbo_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp], _i),
    "ddpm_elbo_tanh_head": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_time_embedding": ([_vp, _f, _i, _vp, _i, _vp], _i),
    "ddpm_nchw_f32_to_nhwc_bf16": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_nhwc_bf16_to_nchw_f32": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_tanh_bwd_nchw_to_nhwc": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_bump_seed": ([_vp, _vp], _i),
    "ddpm_check_timesteps": ([_vp, _vp, _i, _i, _i, _vp, _vp], _i),
    "ddpm_sample_timesteps": ([_vp, _i, _i, _vp, _u32, _vp], _i),
    "ddpm_act_backward": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "ddpm_prepare_images": ([_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp], _i),
    "ddpm_conv_forward": ([C.POINTER(ConvDesc), _vp], _i),
    "ddpm_conv_debug_tile": ([_i], _i),
    "ddpm_conv_last_kernel": ([], _i),
    "ddpm_conv_pipe_mode": ([_i], _i),
    "ddpm_conv_pw_stream_mode": ([_i], _i),
    "ddpm_conv_gn_fusable": ([C.POINTER(ConvDesc)], _i),
    "ddpm_conv_gn_part_tiles": ([C.POINTER(ConvDesc)], _i),
    "ddpm_conv_gnb_fusable": ([C.POINTER(ConvDesc)], _i),
    "ddpm_gn_act_forward": ([C.POINTER(GnDesc), _vp], _i),
    "ddpm_gn_act_backward": ([C.POINTER(GnDesc), _vp], _i),
    "ddpm_gn_scratch_floats": ([C.POINTER(GnDesc), _i], _i64),
    "ddpm_gn_fold_affine": ([_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp], _i),
    "ddpm_gn_act_forward_static": ([C.POINTER(GnDesc), _vp], _i),
    "ddpm_gn_static_scratch_floats": ([], _i64),
    "ddpm_bn_update_running": ([_vp, _vp, _vp, _vp, _i, _i64, _f, _f, _vp], _i),
    "ddpm_bn_eval_stats": ([_vp, _vp, _vp, _i, _f, _vp], _i),
    "ddpm_dwconv3x3_forward": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_dwconv3x3_stat_parts": ([_i, _i, _i, _i, _i], _i),
    "ddpm_dwconv3x3_forward_stats": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "ddpm_dwconv3x3_affine_supported": ([_i, _i, _i, _i, _i], _i),
    "ddpm_dwconv3x3_forward_affine": ([_vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp], _i),
    "ddpm_dwconv3x3_wgrad": ([_vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "ddpm_dwconv3x3_wgrad_blocks": ([_i, _i, _i, _i], _i),
    "ddpm_reduce_rows": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "ddpm_remap": ([_vp, _vp, _i, _i, _vp], _i),
    "ddpm_remap_chunk_elems": ([], _i),
    "ddpm_conv_wgrad_splits": ([_vp], _i),
    "ddpm_conv_wgrad_pipe_splits": ([_vp], _i),
    "ddpm_conv_wgrad_group_bytes": ([_i], _i64),
    "ddpm_conv_wgrad_group_plan": ([_vp, _i, _vp, _i64], _i),
    "ddpm_conv_wgrad_group_launch": ([_vp, _vp, _vp], _i),
    "ddpm_embed_forward": ([_vp, _vp], _i),
    "ddpm_embed_backward": ([_vp, _vp], _i),
    "ddpm_sir_supported": ([_i, _i, _i, _i, _i, _i], _i),
    "ddpm_sir_backward_supported": ([_i, _i, _i, _i, _i, _i], _i),
    "ddpm_sir_forward": ([_vp, _vp], _i),
    "ddpm_sir_workgroups": ([_i, _i], _i),
    "ddpm_sir_backward": ([_vp, _vp], _i),
    "ddpm_adam_step": ([_vp, _vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _f, _vp, _vp], _i),
    "ddpm_adam_step_range": ([_vp, _vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _f, _vp, _i, _vp], _i),
    "ddpm_conv_wgrad": ([C.POINTER(WgradDesc), _vp], _i),
    "ddpm_colsum": ([_vp, _vp, _i64, _i, _i, _vp], _i),
    "ddpm_spatial_sum": ([_vp, _vp, _i, _i, _i, _vp], _i),
    "ddpm_attention_forward": ([_vp, _vp], _i),
    "ddpm_attention_backward": ([_vp, _vp], _i),
    "ddpm_sample_norm_forward": ([_vp, _vp], _i),
    "ddpm_sample_norm_backward": ([_vp, _vp], _i),
    "ddpm_dropout_masks": ([_vp, _vp, _i, _i64, _f, _vp, _vp], _i),
}

_lib = None


def lib():
    """The loaded library; raises if it has not been built (python diffusion-model_amd/build.py)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build the HIP extension first (python diffusion-model_amd/build.py); "
                               "there is no fallback path")
        _lib = C.CDLL(LIB_PATH)
        _lib.ddpm_last_error.restype = C.c_char_p
        for name, (args, res) in _PROTOS.items():
            fn = getattr(_lib, name)
            fn.argtypes, fn.restype = args, res
        if os.environ.get("DDPM_CONV_PW_STREAM", "1") == "0":     # A/B switch: never take the streaming pointwise kernel
            _lib.ddpm_conv_pw_stream_mode(-1)
        if os.environ.get("DDPM_CONV_PIPE", "1") == "0":     # A/B switch: never take the pipelined wide-convolution kernel
            _lib.ddpm_conv_pipe_mode(-1)                      # (per thread: this one launches the forward passes)
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise RuntimeError(f"libddpm_hip {what} failed ({rc}): {lib().ddpm_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream() -> int:
    """The hipStream_t torch currently enqueues on (so launches are captured by torch.cuda graphs too)."""
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args), name)
