"""ctypes binding of libcrossscore_hip.so (include/crossscore_hip.h).  Fails loudly when the library is
missing: there is no CPU or eager fallback for the hot path."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcrossscore_hip.so")

CS_OK, CS_ERR_BAD_ARG, CS_ERR_UNSUPPORTED, CS_ERR_STATE, CS_ERR_HIP = range(5)

# CsEpilogue (csrc/cs_common.h)
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2  # cs_set_weight_typed
(EPI_BIAS_F16, EPI_BIAS_GELU_F16, EPI_BIAS_RELU_F16, EPI_BIAS_LEAKY_F16, EPI_RESID_F32, EPI_PATCH_F32, EPI_HEAD_SCORE,
 EPI_LN_F16, EPI_LN_GELU_F16, EPI_RESID_F32_LN) = range(10)


class CsConfig(C.Structure):
    _fields_ = [
        ("hidden", C.c_int), ("enc_layers", C.c_int), ("enc_heads", C.c_int), ("mlp_ratio", C.c_int),
        ("patch", C.c_int), ("pos_grid", C.c_int), ("pe_h", C.c_int), ("pe_w", C.c_int),
        ("dec_layers", C.c_int), ("dec_heads", C.c_int), ("do_self_attn", C.c_int), ("do_short_cut", C.c_int),
        ("act", C.c_int), ("pow_p", C.c_float), ("enc_chunk_images", C.c_int), ("ln_fold", C.c_int), ("lanes", C.c_int), ("pos_interp_legacy", C.c_int), ("enc_fused", C.c_int),
        ("operand_dtype", C.c_int), ("pe_interp_mode", C.c_int), ("skip_finite_check", C.c_int), ("swiglu", C.c_int),
    ]


class CsU8Image(C.Structure):
    """cs_u8_image: one decoded uint8 HWC image on the device and its resize / crop geometry (cs_forward_u8)."""
    _fields_ = [("data", C.c_void_p), ("h", C.c_int), ("w", C.c_int), ("row_bytes", C.c_int), ("rs_h", C.c_int), ("rs_w", C.c_int),
                ("crop_y", C.c_int), ("crop_x", C.c_int)]


# every symbol include/crossscore_hip.h declares: name -> (restype, argtypes)
_vp, _i, _f, _ll, _sz = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_size_t
_fp = C.POINTER(C.c_float)
SYMBOLS = {
    "cs_create": (_vp, [C.POINTER(CsConfig)]),
    "cs_destroy": (None, [_vp]),
    "cs_last_error": (C.c_char_p, []),
    "cs_set_weight": (_i, [_vp, C.c_char_p, _vp, _i, _i, C.POINTER(C.c_int64)]),
    "cs_set_weight_typed": (_i, [_vp, C.c_char_p, _vp, _i, _i, _i, C.POINTER(C.c_int64)]),
    "cs_num_weights": (_i, [_vp]),
    "cs_weight_name": (C.c_char_p, [_vp, _i]),
    "cs_finalize": (_i, [_vp]),
    "cs_forward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "cs_encode_references": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "cs_forward_cached": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp]),
    "cs_forward_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _fp, _fp, _vp, _vp, _i, _vp, _vp]),
    "cs_encode_references_u8": (_i, [_vp, _vp, _i, _i, _i, _fp, _fp, _vp, _vp]),
    "cs_forward_cached_u8": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _fp, _fp, _vp, _vp, _i, _vp, _vp]),
    "cs_u8_input_supported": (_i, [_vp, _vp, _i, _i]),
    "cs_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "cs_nonfinite_count": (_i, [_vp, C.POINTER(C.c_longlong)]),
    "cs_forward_stats": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_char_p, _sz]),
    "cs_debug_set_op_operand_dtype": (_i, [_i]),
    "cs_debug_capture": (_i, [_vp, _i]),
    "cs_debug_read": (_i, [_vp, C.c_char_p, _vp, _sz, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), _vp]),
    "cs_profile_enable": (_i, [_vp, _i]),
    "cs_profile_read": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "cs_profile_read_bytes": (_i, [_vp, _i, C.POINTER(C.c_double)]),
    "cs_op_gemm": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _f,
                        _vp, _vp, _i, _vp, _i, _vp, _f, _vp]),
    "cs_op_head_score": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    "cs_op_attention": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _ll, _ll, _ll, _ll, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "cs_op_attention_weights": (_i, [_vp, _vp, _i, _i, _ll, _ll, _i, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp]),
    "cs_op_layernorm": (_i, [_vp, _i, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "cs_op_ln_finalize": (_i, [_vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "cs_op_im2col": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "cs_op_patch_embed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cs_op_patch_embed_fused": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cs_op_patch_embed_fused_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "cs_debug_patch_fused_enable": (None, [_i]),
    "cs_set_lanes": (_i, [_vp, _i]),
    "cs_redraw_lane_streams": (_i, [_vp]),
    "cs_debug_stream_probe_log": (None, [_i]),
    "cs_op_pos_bicubic": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "cs_op_pos_bicubic_ex": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cs_op_pe_bilinear": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "cs_op_pe_interp": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "cs_op_score_to_gray16": (_i, [_vp, C.c_longlong, _i, _vp, _vp]),
    "cs_op_score_to_rgb": (_i, [_vp, C.c_longlong, C.c_float, C.c_float, _vp, _vp, _vp]),
    "cs_op_preprocess_u8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp, _vp]),
    "cs_op_pack_f16": (_i, [_vp, _i, _i, _vp, _i, _vp, _vp, _vp]),
    "cs_op_streams_overlap": (_i, [_vp, _vp, _vp]),
    "cs_op_ln_fold_consts": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "cs_gemm_column_tiles": (_i, [_i]),
    "cs_debug_gemm256_enable": (None, [_i]),
    "cs_debug_gemm256_kmin": (None, [_i]),
    "cs_debug_panel_impl": (None, [_i]),
    "cs_panel_supported": (_i, [_i, _i]),
    "cs_panel_image_bytes": (_sz, [_i]),
    "cs_op_panel_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cs_op_linear_layernorm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _i, _vp]),
    "cs_op_linear_layernorm_linear": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "cs_debug_rowln_enable": (None, [_i]),
    "cs_op_encoder_panel": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
}

_lib = None


class CrossScoreHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the in-tree library and bind every declared symbol.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CrossScoreHipError(
            f"{LIB_PATH} is missing: build it with `python -m crossscore_amd.build` (hipcc --offload-arch=gfx950). "
            "The CrossScore hot path has no CPU/eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().cs_last_error().decode("utf-8", "replace")


def check(rc: int) -> None:
    """Maps library status codes to the exceptions the reference raises for the same conditions:
    ValueError for bad shapes / configs (regression_layer.py:37, check_config.py:23-28, image.py:15 assert),
    RuntimeError for state and HIP errors (load_state_dict / CUDA errors)."""
    if rc == CS_OK:
        return
    msg = last_error()
    if rc in (CS_ERR_BAD_ARG,):
        raise ValueError(msg)
    if rc == CS_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise CrossScoreHipError(msg)
