"""ctypes binding of ``libe4s_hip.so`` (the C ABI declared in ``include/e4s_hip.h``).

The library is built ahead of time (``python -m e4s2024_amd.build`` / ``__graft_entry__.build()``), never
at import.  There is no CPU fallback: if the shared object is missing or a symbol cannot be resolved the
import fails loudly, and every wrapper raises ``RuntimeError`` with the library's message on a non-zero status.
"""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("E4S_HIP_LIB") or os.path.join(HERE, "lib", "libe4s_hip.so")   # E4S_HIP_LIB: tuning builds only
HEADER = os.path.join(os.path.dirname(HERE), "include", "e4s_hip.h")

c_int, c_i64, c_f32, c_ptr = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p

# name -> argtypes; every function returns int status except the two noted below
_PROTOS = {
    "e4s_fused_bias_act": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_f32, c_f32, c_i64, c_i64, c_i64, c_ptr],
    "e4s_upfirdn2d": [c_ptr, c_ptr, c_ptr] + [c_int] * 13 + [c_ptr],
    "e4s_onehot_to_labels": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_modconv_prep_weights": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_style_demod": [c_ptr, c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_style_demod_batched": [c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_region_modconv3x3": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr, c_int] + [c_int] * 7 + [c_ptr, c_i64, c_ptr],
    "e4s_modconv_prep_weights_sb": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_region_modconv3x3_sb": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr, c_int] + [c_int] * 7 + [c_ptr, c_i64] + [c_ptr] * 10,
    "e4s_modconv_mx_weight_bytes": [c_int, c_int, c_int, c_int, c_ptr],
    "e4s_modconv_mx4_weight_bytes": [c_int, c_int, c_ptr],
    "e4s_modconv_prep_weights_mx4": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr],
    "e4s_region_upconv_mx4": [c_ptr] * 8 + [c_int, c_int, c_ptr, c_int, c_ptr, c_ptr] + [c_int] * 7 + [c_ptr],
    "e4s_modconv_prep_weights_mx": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_region_modconv3x3_mx": [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_int, c_ptr, c_ptr, c_int] + [c_int] * 7 + [c_ptr, c_i64] + [c_ptr] * 10,
    "e4s_upblock_mx_weight_bytes": [c_int, c_int, c_ptr],
    "e4s_modconv_prep_weights_upblock_mx": [c_ptr, c_ptr, c_int, c_int, c_ptr],
    "e4s_masked_upconv_blocks_mx": [c_ptr] * 9 + [c_ptr, c_int, c_ptr, c_ptr] + [c_int] * 7 + [c_ptr],
    "e4s_conv_prep_weights_mx": [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_conv3x3_mx": [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv3x3_mx3_weight_bytes": [c_int, c_int, c_ptr],
    "e4s_conv_prep_weights_mx3": [c_ptr, c_ptr, c_int, c_int, c_ptr],
    "e4s_conv3x3_mx3": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv_prep_weights_mx3_s2": [c_ptr, c_ptr, c_int, c_int, c_ptr],
    "e4s_conv3x3_s2_mx3": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv3x3_mx3_phased": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv3x3_mx3_ex": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_uniform_blocks": [c_ptr, c_ptr, c_ptr, c_ptr] + [c_int] * 8 + [c_ptr],
    "e4s_modconv_tconv_sb": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_modconv_up_fused_sb": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_int, c_ptr, c_ptr],
    "e4s_swap_head_mask": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_foreground_masks": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_frames_to_tensor": [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_erode_labels": [c_ptr, c_ptr, c_int, c_int, c_int, c_int, ctypes.c_uint, c_ptr],
    "e4s_pyr_down": [c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_resample_u8": [c_ptr] * 5 + [c_int] * 7 + [c_ptr],
    "e4s_pyr_blend_level": [c_ptr] * 7 + [c_int, c_int, c_int, c_ptr],
    "e4s_pyr_up": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_mconv_unfold": [c_ptr] * 4 + [c_int] * 7 + [c_ptr],
    "e4s_mconv_scale": [c_ptr] * 9 + [c_int, c_ptr, c_ptr] + [c_int] * 8 + [c_ptr],
    "e4s_style_tables_bwd": [c_ptr] * 14 + [c_f32] * 3 + [c_int] * 5 + [c_ptr],
    "e4s_unfold2d": [c_ptr, c_ptr] + [c_int] * 9 + [c_ptr],
    "e4s_mconv_fold": [c_ptr] * 6 + [c_int] * 8 + [c_ptr],
    "e4s_gemm_sb": [c_ptr] * 3 + [c_int] * 7 + [c_i64] * 3 + [c_int, c_ptr, c_i64, c_ptr],
    "e4s_mconv_wgrad": [c_ptr] * 5 + [c_int] * 8 + [c_ptr, c_i64, c_ptr],
    "e4s_wino_weight": [c_ptr, c_ptr, c_int, c_int, c_ptr],
    "e4s_wino_input": [c_ptr] * 4 + [c_int] * 4 + [c_ptr],
    "e4s_wino_output": [c_ptr] * 3 + [c_int] * 4 + [c_ptr],
    "e4s_mconv_dgrad_tiles": [c_int] * 2,
    "e4s_mconv_dgrad": [c_ptr] * 7 + [c_int] * 7 + [c_ptr],
    "e4s_blur_epilogue": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_region_torgb": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr, c_ptr, c_ptr] + [c_int] * 5 + [c_ptr],
    "e4s_conv_prep_weights": [c_ptr] * 7 + [c_f32, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv_prep_weights_f16x3": [c_ptr] * 8 + [c_f32, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv2d_f16x3": [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int] + [c_int] * 9 + [c_ptr, c_ptr],
    "e4s_conv2d": [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int] + [c_int] * 8 + [c_ptr],
    "e4s_conv_prep_weights_sb": [c_ptr] * 8 + [c_f32, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv_prep_weights_sb3": [c_ptr] * 9 + [c_f32, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_conv2d_sb": [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int] + [c_int] * 8 + [c_ptr],
    "e4s_conv2d_sb3": [c_ptr, c_ptr, c_ptr, c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_int] + [c_int] * 8 + [c_ptr],
    "e4s_plane_stats": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_f32, c_ptr],
    "e4s_vec_fc": [c_ptr] * 7 + [c_f32, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_norm_gate_add": [c_ptr] * 8 + [c_int, c_ptr, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_se_gate": [c_ptr] * 4 + [c_int] * 3 + [c_ptr],
    "e4s_norm_gate_add_stats": [c_ptr] * 10 + [c_int, c_ptr, c_int, c_int, c_int, c_int, c_f32, c_ptr],
    "e4s_norm_self_gate_add_stats": [c_ptr] * 4 + [c_f32] + [c_ptr] * 4 + [c_int, c_ptr, c_int, c_int, c_int, c_int, c_f32, c_ptr],
    "e4s_masked_avg_pool": [c_ptr, c_ptr, c_ptr] + [c_int] * 7 + [c_ptr],
    "e4s_bilinear_resize": [c_ptr, c_ptr] + [c_int] * 6 + [c_ptr],
    "e4s_maxpool3x3s2": [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_gate_add_upsample": [c_ptr] * 5 + [c_int] * 4 + [c_ptr],
    "e4s_bilinear_argmax": [c_ptr, c_ptr, c_ptr] + [c_int] * 6 + [c_ptr],
    "e4s_conv7x7s2_stem_f16x3": [c_ptr] * 4 + [c_int] * 5 + [c_ptr],
    "e4s_bicubic_down_normalize": [c_ptr] * 5 + [c_int] * 5 + [c_ptr],
    "e4s_bicubic_down_normalize_pm1": [c_ptr] * 5 + [c_int] * 5 + [c_ptr],
    "e4s_tensor2im_u8": [c_ptr, c_ptr, c_int, c_int, c_int, c_ptr],
    "e4s_to_split_planes": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_int, c_int, c_int, c_ptr],
    "e4s_chain_conv3x3": [c_ptr, c_ptr],
    "e4s_modconv_prep_weights_hc": [c_ptr, c_ptr, c_ptr, c_ptr, c_int, c_int, c_ptr],
    "e4s_modconv_up_hc": [c_ptr] * 7 + [c_int, c_ptr, c_ptr, c_int] + [c_int] * 5 + [c_ptr, c_ptr],
    "e4s_small_map": [c_ptr, c_ptr, c_ptr, c_int, c_int, c_i64, c_int, c_int, c_ptr],
    "e4s_grouped_linear_bwd": [c_ptr] * 8 + [c_f32, c_f32, c_f32] + [c_int] * 5 + [c_ptr],
    "e4s_grouped_linear": [c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_int, c_f32] + [c_int] * 4 + [c_ptr],
}


def declared_symbols(header: str = HEADER):
    """Names of every entry point declared in include/e4s_hip.h."""
    with open(header) as f:
        return re.findall(r"E4S_API\s+[\w\s\*]+?\b(e4s_\w+)\s*\(", f.read())


class StyleJob(ctypes.Structure):
    """Mirror of E4sStyleJob (include/e4s_hip.h)."""
    _fields_ = [("s", c_ptr), ("d", c_ptr), ("styles", c_ptr), ("stride_b", c_i64), ("stride_r", c_i64), ("mod_weight", c_ptr),
                ("mod_bias", c_ptr), ("wsq", c_ptr), ("nreg", c_int), ("cin", c_int), ("cout", c_int), ("_pad", c_int)]


class ChainLayer(ctypes.Structure):
    """Mirror of E4sChainLayer (include/e4s_hip.h)."""
    _fields_ = [(n, c_ptr) for n in ("x_sp", "whi", "wlo", "d", "noise", "noise_weight", "act_bias", "out_sp", "s_next", "rgb_out", "rgb_wt",
                                      "rgb_s", "rgb_bias", "rgb_skip", "rgb_up_kernel")] + \
               [(n, c_int) for n in ("noise_bs", "act", "bs", "cin", "cout", "h", "w", "_pad")]


class _Lib:
    def __init__(self):
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} not found: build the HIP library first (python -m e4s2024_amd.build). "
                "e4s2024_amd has no CPU fallback.")
        # torch first: its wheel carries its own libamdhip64, and the process must end up with ONE HIP runtime — the library loaded before torch
        # brings in /opt/rocm's copy, torch then its own, and the kernels launch into a runtime that has no device ("no ROCm-capable device")
        import torch  # noqa: F401
        self.cdll = ctypes.CDLL(SO_PATH)
        self.cdll.e4s_last_error.restype = ctypes.c_char_p
        self.cdll.e4s_last_error.argtypes = []
        self.cdll.e4s_abi_version.restype = c_int
        self.cdll.e4s_abi_version.argtypes = []
        for name, args in _PROTOS.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so is stale
            fn.argtypes = args
            fn.restype = c_int
        self.path = SO_PATH

    def call(self, name: str, *args):
        st = getattr(self.cdll, name)(*args)
        if st != 0:
            raise RuntimeError(f"{name} failed (status {st}): {self.cdll.e4s_last_error().decode()}")


_lib = None


def lib() -> _Lib:
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib
