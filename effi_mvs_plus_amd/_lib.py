"""ctypes binding of the C-ABI kernel library (``include/effi_mvs_hip.h``).

The library is built in-tree by ``make -C effi_mvs_plus_amd/csrc`` (see ``__graft_entry__.build``).
There is deliberately no fallback: if the shared object is missing, ``lib()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EFFI_MVS_LIB") or os.path.join(_HERE, "libeffimvs_hip.so")   # override: A/B runs of two builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "effi_mvs_hip.h")

_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float

# name -> argtypes (restype is int for all but effi_error_string)
SIGNATURES = {
    "effi_version": [],
    "effi_set_workspace": [_i, _vp, _l],
    "effi_set_option": [C.c_char_p, _l],
    "effi_fusion_dynamic_filter_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _f, _i, _f, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp],
    "effi_fusion_dtu_filter_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _f, _f, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "effi_fusion_vis_filter_f32": [_vp, _vp, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp],
    "effi_fusion_points_f32": [_i, _vp, _l, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "effi_fusion_dtu_reproject_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp],
    "effi_compose_rel_proj_f32": [_vp, _i, _vp, _vp],
    "effi_rel_proj_f32": [_vp, _vp, _vp, _vp],
    "effi_planar_to_nhwc_f32": [_vp, _vp, _i, _i, _i, _vp],
    "effi_homo_warp_f32": [_vp, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp],
    "effi_warpcorr_views_f32": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_warpcorr_views_tbl_f32": [_vp, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_warpcorr_views_x3_f32": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "effi_warpcorr_views_x3_tbl_f32": [_vp, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp, _i, _vp],
    "effi_warpcorr_dyn_tbl_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_view_table_set": [_vp, _vp, _i, _vp],
    "effi_pixelwise_net_f32": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "effi_view_aggregate_f32": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "effi_warpcorr_dyn_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_conv3d_k3_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_deconv3d_k3s2_bf16x3_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_conv3d_k3s2_mfma_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_conv3d_k3s1_mfma_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_deconv3d_k3_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_softmax_regress_conf_f32": [_vp, _vp, _l, _l, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "effi_vol_lookup1d_f32": [_vp, _l, _l, _i, _vp, _l, _l, _l, _i, _vp, _vp, _l, _i, _i, _vp, _vp],
    "effi_bilinear_sampler1d_f32": [_vp, _i, _i, _i, _vp, _l, _vp, _vp, _vp],
    "effi_getcost_conv1x1_f32": [_vp, _vp, _i, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _i,
                                 _i, _vp, _vp],
    "effi_getcost_f32": [_vp, _vp, _i, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp],
    "effi_conv2d_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "effi_conv3d_k3s1_roll_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_conv3d_k3s1_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_vol_lookup1d_pair_f32": [_vp, _vp, _l, _l, _i, _vp, _l, _l, _l, _i, _vp, _vp, _l, _i, _i, _vp, _vp, _vp],
    "effi_conv3d_k3_pair_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "effi_conv3d_k3s1_roll_bf16x3_pair_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "effi_csp_gen_roll_bf16x3_pair_f32": [_vp, _i, _i, _i] + [_vp] * 16 + [_vp],
    "effi_debug_poison_lds": [C.c_uint, _vp, _vp],
    "effi_debug_pk_war_probe": [_vp, _i, _i, _i, _vp],
    "effi_deconv3d_k3_pair_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "effi_homo_warp_bwd_f32": [_vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_warpcorr_views_bwd_f32": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "effi_image_prepare_u8_f32": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_resize_linear_f32": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_encoder_inputs_f32": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp,
                                _i, _vp, _vp, _vp],
    "effi_encoder_inputs_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp,
                                       _i, _vp, _vp, _vp],
    "effi_conv2d_k3_bf16x3_pair_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "effi_conv2d_k3_k1_up2x_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "effi_conv2d_k3_k1_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "effi_conv2d_k3_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "effi_conv2d_k5s2_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "effi_conv2d_k5s2_bf16x3_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "effi_conv3d_k3s2_bf16x3_f32": [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_conv2d_c1k7_relu_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "effi_conv2d_c1k7_relu_bf16x3_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "effi_convex_upsample2x_f32": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "effi_compose_rel_proj_stages_f32": [_vp, _i, _i, _vp, _vp],
    "effi_split_tanh_relu_stages_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "effi_split_tanh_relu_f32": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "effi_depth_to_inv_f32": [_vp, _vp, _i, _i, _vp, _vp],
    "effi_stage1_hypotheses_f32": [_vp, _i, _i, _vp, _vp, _vp],
    "effi_upsample_nearest_f32": [_vp, _i, _i, _i, _i, _vp, _vp],
    "effi_head_update_f32": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "effi_conv2d_k3_twice_bf16x3_f32": [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "effi_encoder_tail_bf16x3_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp],
    # split-resident maps of the update block
    "effi_sr_geometry": [_i, _i, _vp, _vp],
    "effi_sr_clear_border": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "effi_sr_from_planar_f32": [_vp, _i, _i, _i, _vp, _i, _i, _vp],
    "effi_split_tanh_relu_stages_sr_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "effi_encoder_inputs_bf16x3_sr": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp,
                                      _i, _vp, _vp, _i, _i, _vp],
    "effi_encoder_pair_gen_bf16x3_sr": [_vp, _vp, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp, _vp, _vp,
                                        _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "effi_gru_zr_q_fused_bf16x3_sr": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_conv2d_k3_bf16x3_sr": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "effi_conv2d_k3_bf16x3_pair_sr": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "effi_conv2d_k3_k1_bf16x3_sr": [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_conv2d_k3_k1_up2x_bf16x3_sr": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    # scope row n2: training kernels
    "effi_conv_wgrad_f32": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp],
    "effi_channel_sum_f32": [_vp, _i, _i, _l, _vp, _vp, _i, _vp],
    "effi_conv2d_k5s2_dgrad_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "effi_bn_moment_f32": [_vp, _i, _i, _l, _vp, _i, _vp, _vp, _i, _vp],
    "effi_bn_train_fwd_f32": [_vp, _i, _i, _l, _vp, _vp, _f, _f, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "effi_pack_conv2d_mfma_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_pack_conv2d_k5s2_dgrad_f32": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "effi_bn_apply_f32": [_vp, _i, _i, _l, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "effi_bn_bwd_f32": [_vp, _vp, _vp, _i, _i, _l, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "effi_pointwise_f32": [_i, _vp, _vp, _vp, _vp, _f, _f, _l, _l, _i, _vp, _vp, _vp, _vp],
    "effi_vol_lookup1d_bwd_f32": [_vp, _l, _l, _i, _vp, _l, _l, _l, _i, _vp, _vp, _l, _i, _i, _vp, _vp],
    "effi_getcost_bwd_f32": [_vp, _vp, _i, _i, _vp, _vp, _l, _l, _i, _vp, _l, _l, _i, _vp, _vp, _l, _i, _i, _i, _vp, _vp],
    "effi_softargmin_bwd_f32": [_vp, _vp, _l, _l, _i, _i, _vp, _vp, _vp],
    "effi_view_aggregate_bwd_f32": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "effi_convex_upsample2x_bwd_f32": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "effi_warpcorr_dyn_bwd_f32": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
}

# plain-bf16-operand variants (hi*hi only) of the split-precision convolution entries: same signatures, suffix _bf16
BF16X3_ENTRIES = ("effi_conv2d_k3_bf16x3_pair_f32", "effi_conv2d_k3_bf16x3_f32", "effi_conv2d_k3_k1_bf16x3_f32",
                  "effi_conv2d_k3_k1_up2x_bf16x3_f32", "effi_conv3d_k3s1_bf16x3_f32", "effi_conv3d_k3s1_roll_bf16x3_f32",
                  "effi_conv3d_k3s1_roll_bf16x3_pair_f32", "effi_csp_gen_roll_bf16x3_pair_f32", "effi_deconv3d_k3s2_bf16x3_f32", "effi_encoder_tail_bf16x3_f32", "effi_conv2d_k3_twice_bf16x3_f32",
                  "effi_conv2d_k5s2_bf16x3_f32", "effi_conv3d_k3s2_bf16x3_f32",
                  "effi_conv2d_k3_bf16x3_sr", "effi_conv2d_k3_bf16x3_pair_sr", "effi_conv2d_k3_k1_bf16x3_sr", "effi_conv2d_k3_k1_up2x_bf16x3_sr",
                  "effi_encoder_pair_gen_bf16x3_sr", "effi_gru_zr_q_fused_bf16x3_sr")
for _n in BF16X3_ENTRIES:
    SIGNATURES[_n + "_bf16"] = SIGNATURES[_n]

# entry points whose return type is not the int status code (bound explicitly in lib())
NON_STATUS_SYMBOLS = ("effi_error_string", "effi_workspace_bytes", "effi_get_workspace", "effi_get_option", "effi_option_unset")

_lock = threading.Lock()
_lib = None


class EffiLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise if the HIP library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise EffiLibraryError(
                f"{LIB_PATH} not found: build the gfx950 kernels first "
                "(python -c 'import __graft_entry__ as g; g.build()' or make -C effi_mvs_plus_amd/csrc). "
                "There is no CPU / eager fallback for the cost-volume path.")
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError here = header/library mismatch
            fn.argtypes = argtypes
            fn.restype = _i
        handle.effi_error_string.argtypes = [_i]
        handle.effi_error_string.restype = C.c_char_p
        handle.effi_workspace_bytes.argtypes = []
        handle.effi_workspace_bytes.restype = _l
        handle.effi_get_workspace.argtypes = [_i]
        handle.effi_get_workspace.restype = _vp
        handle.effi_get_option.argtypes = [C.c_char_p]
        handle.effi_get_option.restype = _l
        handle.effi_option_unset.argtypes = []
        handle.effi_option_unset.restype = _l
        _lib = handle
    return _lib


def check(code: int, what: str):
    if code != 0:
        msg = lib().effi_error_string(code).decode()
        raise EffiLibraryError(f"{what} failed with code {code}: {msg}")


def declared_symbols():
    """Names of every function the public header declares (used by the CPU-side export test)."""
    import re
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(effi_[a-z0-9_]+)\s*\(", text)))
