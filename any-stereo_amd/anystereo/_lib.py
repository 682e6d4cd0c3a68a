"""ctypes binding of libanystereo_hip.so (C ABI: include/anystereo_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails the product path
raises.  (The CPU oracle lives under /oracle and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own libamdhip64.so.7; it MUST be the HIP runtime this library binds to, otherwise
# stream handles taken from torch would belong to a different runtime instance (and loading /opt/rocm's
# copy first makes torch.cuda unusable).  Importing torch first pins the shared runtime.
import torch  # noqa: F401

AS_MAX_LEVELS = 4
AS_MAX_SRCS = 4
AS_F32, AS_F16, AS_F64 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_RELU6, ACT_LEAKY, ACT_GELU = 0, 1, 2, 3, 4, 5, 6
EPI_LINEAR, EPI_GRU_ZR, EPI_GRU_Q, EPI_RELU_TAPS = 0, 1, 2, 4

# ANYSTEREO_LIB selects another build of the same library (A/B timing of kernel variants); default = the in-tree build
LIB_PATH = os.environ.get("ANYSTEREO_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libanystereo_hip.so")

_vp, _i, _fp = C.c_void_p, C.c_int, C.c_void_p
_pp = C.POINTER(C.c_void_p)


class ConvDesc(C.Structure):
    """as_conv_desc (include/anystereo_hip.h)."""
    _fields_ = [
        ("src", C.c_void_p * AS_MAX_SRCS), ("src_c", C.c_int * AS_MAX_SRCS), ("n_src", C.c_int),
        ("wpack", C.c_void_p), ("bias", C.c_void_p), ("add", C.c_void_p),
        ("add_ctot", C.c_int), ("add_coff", C.c_int),
        ("h", C.c_void_p), ("z", C.c_void_p), ("out", C.c_void_p), ("out2", C.c_void_p),
        ("out_ctot", C.c_int), ("out_coff", C.c_int),
        ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("KS", C.c_int),
        ("act", C.c_int), ("epilogue", C.c_int), ("precision", C.c_int),
        ("ws", C.c_void_p), ("ws_elems", C.c_int64), ("stride", C.c_int),
        ("src_bs", C.c_int * AS_MAX_SRCS), ("out_bs", C.c_void_p), ("out_bs_ctot", C.c_int), ("out_bs_coff", C.c_int),
        ("bs_only", C.c_int),
        ("tap_w", C.c_void_p), ("dual", C.c_int), ("src2", C.c_void_p), ("src2_bs", C.c_int), ("wpack2", C.c_void_p), ("bias2", C.c_void_p),
        ("out_coff2", C.c_int), ("out_bs_coff2", C.c_int),
        ("h2", C.c_void_p), ("dual_act2", C.c_int), ("act2", C.c_int), ("out_b", C.c_void_p), ("out_bs_b", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/anystereo_hip.h declares
SIGNATURES = {
    "as_set_precision": (_i, [_i]),
    "as_get_precision": (_i, []),
    "as_set_fast16": (_i, [_i]),
    "as_get_fast16": (_i, []),
    "as_last_error_string": (C.c_char_p, []),
    "as_abi_version": (_i, []),
    "as_graph_replace_memsets": (_i, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "as_stamp": (_i, [_vp, _i, _vp]),
    "as_ir_block_pack_bytes": (C.c_int64, [_i, _i, _i]),
    "as_ir_block_pack": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "as_ir_block": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_source_hash": (C.c_char_p, []),
    "as_device_count": (_i, []),
    "as_corr_sampler_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_corr_sampler_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_corr_build_pyramid": (_i, [_vp, _vp, _pp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_geo_pyramid": (_i, [_vp, _pp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_geo_corr_lookup_fwd": (_i, [_pp, _pp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_geo_corr_lookup_bwd": (_i, [_vp, _vp, _pp, _pp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_geo_corr_lookup_bwd_accum": (_i, [_vp, _vp, _pp, _pp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_lookup_convc1_pack_bytes": (C.c_int64, [_i]),
    "as_lookup_convc1_pack": (_i, [_vp, _i, _vp, _vp]),
    "as_lookup_convc1_fwd": (_i, [_pp, _pp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_loop_front_fwd": (_i, [_pp, _pp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i,
                                _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_lookup_split_overflow": (C.c_uint, [_i]),
    "as_conv_split_overflow": (C.c_uint, [_i]),
    "as_volumes_split_overflow": (C.c_uint, [_i]),
    "as_gwc_volume_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_disparity_regression": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "as_conv2d": (_i, [C.POINTER(ConvDesc), _vp]),
    "as_conv_ws_elems": (C.c_int64, [_i, _i, _i, _i]),
    "as_conv_pack_size": (C.c_int64, [_i, _i, _i]),
    "as_conv_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "as_conv_pack_size_split": (C.c_int64, [_i, _i, _i]),
    "as_conv_pack_weights_split": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "as_conv_pack_weights_split_t": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "as_conv7x7_c1_relu": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "as_conv3x3_to1": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_tap_shift_sum": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_pool2x": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "as_pool2x_bs": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "as_interp_bilinear_ac_bs": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_dwconv3x3": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_conv3d_k3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_conv3d_k3_gated": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_deconv3d_k4s2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_instance_norm_ws_bytes": (C.c_int64, [_i]),
    "as_instance_norm_act": (_i, [_vp, _vp, _vp, _vp, _i, C.c_int64, C.c_float, _i, _vp]),
    "as_layernorm2d_act": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, C.c_float, _i, _vp]),
    "as_interp_bilinear_ac": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_structure_feature": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_liif_gather": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_liif_gather_mlp1": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_convex_upsample": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "as_conv3x3_few": (_i, [_vp, _vp, _vp, _vp] + [_i] * 7 + [_vp]),
    "as_conv7x7_c3_pack_bytes": (C.c_int64, []),
    "as_conv7x7_c3_pack": (_i, [_vp, _vp, _vp]),
    "as_conv7x7_c3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_liif_latent": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "as_liif_latent_bwd": (_i, [_vp, _vp, _vp] + [_i] * 9 + [_vp]),
    "as_convex_upsample_quater": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "as_convex_upsample_quater_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "as_affinity_bwd": (_i, [_vp, _vp, C.c_int64, _vp, C.c_int64, _vp, C.c_int64, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_liif_affinity": (_i, [_pp, C.POINTER(C.c_int), _i, _vp, _vp, _i, _i, _i, _vp]),
    "as_liif_affinity_ws_bytes": (C.c_int64, [_i, _i, _i, _i]),
    "as_liif_lowres_pack_bytes": (C.c_int64, [_i]),
    "as_liif_lowres_pack": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "as_liif_lowres_cl": (_i, [_pp, C.POINTER(C.c_int), _i, _vp, _vp, _i, _i, _i, _vp]),
    "as_liif_tail_image_bytes": (C.c_int64, []),
    "as_liif_tail_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "as_liif_tail": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "as_liif_query_rows": (_i, [_vp, _i, _i, _vp, _vp]),
    "as_liif_rows_pitch": (_i, []),
    "as_liif_rows_cl": (_i, [_pp, C.POINTER(C.c_int), _i, _vp, _i, _i, _i, _vp]),
    "as_liif_tail_direct": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "as_liif_mlp_bwd_image_bytes": (C.c_int64, []),
    "as_liif_mlp_bwd_pack": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "as_liif_mlp_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_liif_mlp_bwd": (_i, [_vp] * 15 + [_i] * 7 + [_vp]),
    "as_liif_split_overflow": (C.c_uint, [_i]),
    "as_corr_pyramid_bwd": (_i, [_pp, _vp, C.c_longlong, _i, _i, _vp]),
    "as_geo_pyramid_bwd": (_i, [_pp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_conv2d_wgrad_ws_bytes": (C.c_int64, [_i, _i, _i, _i, _i, _i]),
    "as_conv2d_wgrad": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, C.c_int64, _vp]),
    "as_conv2d_wgrad_multi": (_i, [_pp, _pp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_int64, _vp]),
    "as_conv7x7_c1_wgrad_ws_bytes": (C.c_int64, [_i, _i, _i, _i]),
    "as_conv7x7_c1_wgrad_multi": (_i, [_pp, _pp, _i, _i, _vp, _vp, _i, _i, _i, _vp, C.c_int64, _vp]),
    "as_pool2x_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "as_dwconv3x3_s2_bwd_data": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_dwconv3x3_wgrad_slices": (_i, [_i, _i, _i, _i, _i]),
    "as_dwconv3x3_wgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_interp_bilinear_ac_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_gwc_volume_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_disparity_regression_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "as_liif_gather_bwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_liif_gather_bwd_det": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_gru_gates_zr": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_gru_gates_zr_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_gru_gates_q": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_gru_gates_q_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "as_gru_gates_zr_bwd_ctx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_gru_gates_q_bwd_ctx": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "as_liif_rel_key": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "as_convex_upsample_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the library (once) and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libanystereo_hip.so not found at {LIB_PATH}: build it with `python any-stereo_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    from ._srchash import source_hash
    want, have = source_hash(), lib.as_source_hash().decode()
    if want is not None and want != have and os.environ.get("ANYSTEREO_ALLOW_STALE_LIB", "0") != "1":
        raise RuntimeError(f"{LIB_PATH} was built from other sources (library {have}, tree {want}): run `python any-stereo_amd/build.py` "
                           "(ANYSTEREO_ALLOW_STALE_LIB=1 overrides)")
    _lib = lib
    return lib


def library_info() -> dict:
    """What is loaded: path, ABI version, the source hash compiled into it and whether it matches the tree."""
    from ._srchash import source_hash
    lib = load()
    have, want = lib.as_source_hash().decode(), source_hash()
    return {"path": os.path.relpath(LIB_PATH), "abi": int(lib.as_abi_version()), "src_hash": have,
            "matches_sources": None if want is None else have == want}


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().as_last_error_string().decode("utf-8", "replace")
        raise RuntimeError(f"anystereo HIP call {what} failed (code {rc}): {msg}")


def ptr_array(ptrs):
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return C.cast(arr, _pp), arr  # keep `arr` alive for the duration of the call
