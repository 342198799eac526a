"""ctypes binding of libhandnet_hip.so (the C ABI declared in include/handnet_hip.h).

There is no CPU fallback: if the library is missing it is built with hipcc, and if that
fails (or a call returns non-zero) a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

from . import build as _build

ABI_VERSION = 35

TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x64, TILE_128x32, _TILE_RETIRED_5, TILE_64x128 = range(7)
HN_FCOS_MAX_LEVELS = 5

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)


class ConvDesc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "n", "h", "w", "cin", "cout", "r", "s", "stride", "pad", "dil", "oh", "ow",
        "relu_cols", "res_mode", "res_h", "res_w", "in_affine", "tile", "out_split", "res_split",
        "res_pix_stride", "in_pix_stride", "out_pix_stride", "in_affine_stride", "splitk", "terms")]


CONV_MAX_GROUP = 6


class ConvGroup(C.Structure):  # == struct hn_conv_group
    _fields_ = [("count", C.c_int32), ("x16", C.c_void_p * CONV_MAX_GROUP), ("w16", C.c_void_p * CONV_MAX_GROUP),
                ("bias", C.c_void_p * CONV_MAX_GROUP), ("y", C.c_void_p * CONV_MAX_GROUP),
                ("gn_partial", C.c_void_p * CONV_MAX_GROUP), ("h", C.c_int32 * CONV_MAX_GROUP),
                ("w", C.c_int32 * CONV_MAX_GROUP), ("gn_units", C.c_int32)]


CONV_MULTI_MAX = 4


class ConvMulti(C.Structure):  # == struct hn_conv_multi
    _fields_ = [("count", C.c_int32), ("desc", ConvDesc * CONV_MULTI_MAX), ("x16", C.c_void_p * CONV_MULTI_MAX),
                ("w16", C.c_void_p * CONV_MULTI_MAX), ("bias", C.c_void_p * CONV_MULTI_MAX),
                ("residual", C.c_void_p * CONV_MULTI_MAX), ("y", C.c_void_p * CONV_MULTI_MAX)]


class GraphCsr(C.Structure):  # == struct hn_graph_csr
    _fields_ = [("indptr", C.c_void_p), ("indices", C.c_void_p), ("values", C.c_void_p), ("v", C.c_int32)]


class ConvertOpts(C.Structure):  # == struct hn_convert_opts
    _fields_ = [("clamp_keypoints", C.c_int32), ("clamp_box_h", C.c_int32), ("clamp_box_w", C.c_int32), ("reserved", C.c_int32),
                ("sample_box", C.c_void_p), ("sample_paras", C.c_void_p)]


class ModelConfig(C.Structure):  # == struct hn_model_config
    _fields_ = [(k, C.c_int32) for k in ("parts", "num_classes", "num_joints", "rgbd", "min_size", "max_size", "ext", "f16_terms",
                                          "precision")] + [("image_mean", C.c_float * 3), ("image_std", C.c_float * 3)]


MODEL_FCOS, MODEL_A2J = 1, 2
PRECISION_SPLIT, PRECISION_F32 = 0, 1
RANGE_ACTIVATION, RANGE_INPUT, RANGE_INPUT_NONFINITE = 1, 2, 4


class FcosLevels(C.Structure):
    _fields_ = [("num_levels", C.c_int32),
                ("h", C.c_int32 * HN_FCOS_MAX_LEVELS),
                ("w", C.c_int32 * HN_FCOS_MAX_LEVELS),
                ("stride", C.c_int32 * HN_FCOS_MAX_LEVELS),
                ("cls_lr", C.c_void_p * HN_FCOS_MAX_LEVELS),
                ("reg_ctr", C.c_void_p * HN_FCOS_MAX_LEVELS)]


class GnLevels(C.Structure):  # == struct hn_gn_levels
    _fields_ = [("count", C.c_int32), ("hw", C.c_int32 * HN_FCOS_MAX_LEVELS),
                ("partial", C.c_void_p * HN_FCOS_MAX_LEVELS), ("scale", C.c_void_p * HN_FCOS_MAX_LEVELS),
                ("shift", C.c_void_p * HN_FCOS_MAX_LEVELS)]


class SplitLevels(C.Structure):  # == struct hn_split_levels
    _fields_ = [("count", C.c_int32), ("hw", C.c_int32 * HN_FCOS_MAX_LEVELS),
                ("x", C.c_void_p * HN_FCOS_MAX_LEVELS), ("scale", C.c_void_p * HN_FCOS_MAX_LEVELS),
                ("shift", C.c_void_p * HN_FCOS_MAX_LEVELS), ("y16", C.c_void_p * HN_FCOS_MAX_LEVELS)]


class ThinLevels(C.Structure):  # == struct hn_thin_levels
    _fields_ = [("count", C.c_int32), ("x16", C.c_void_p * HN_FCOS_MAX_LEVELS), ("y", C.c_void_p * HN_FCOS_MAX_LEVELS),
                ("h", C.c_int32 * HN_FCOS_MAX_LEVELS), ("w", C.c_int32 * HN_FCOS_MAX_LEVELS)]


class ThinMember(C.Structure):  # == struct hn_thin_member
    _fields_ = [("lv", ThinLevels), ("cout", C.c_int32), ("relu_cols", C.c_int32), ("w16", C.c_void_p), ("bias", C.c_void_p)]


class ThinAffine(C.Structure):  # == struct hn_thin_affine
    _fields_ = [("x", C.c_void_p * HN_FCOS_MAX_LEVELS), ("scale", C.c_void_p * HN_FCOS_MAX_LEVELS),
                ("shift", C.c_void_p * HN_FCOS_MAX_LEVELS), ("in_pix_stride", C.c_int32), ("affine_stride", C.c_int32)]


# name -> (restype, argtypes); every symbol of include/handnet_hip.h is listed here and
# tests/test_abi.py checks the two stay in sync.
VP = C.c_void_p
SIGNATURES = {
    "hn_abi_version": (C.c_int, []),
    "hn_last_error": (C.c_char_p, []),
    "hn_device_info": (C.c_int, [c_i32p, c_i32p, C.c_char_p, C.c_int]),
    "hn_event_create": (C.c_int, [C.POINTER(VP)]),
    "hn_event_destroy": (C.c_int, [VP]),
    "hn_event_record": (C.c_int, [VP, VP]),
    "hn_event_elapsed_ms": (C.c_int, [VP, VP, c_f32p]),
    "hn_clock_sample": (C.c_int, [C.c_int, VP, VP]),
    "hn_conv2d_nhwc_f16x3_multi": (C.c_int, [C.POINTER(ConvMulti), VP, C.c_int64, VP]),
    "hn_conv2d_f16x3_multi_fuses": (C.c_int, [C.POINTER(ConvMulti), C.c_int64]),
    "hn_device_pci_bus_id": (C.c_int, [C.c_char_p, C.c_int]),
    "hn_ingest_u8bgr_u16mm": (C.c_int, [VP, VP, C.c_int, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "hn_range_check_enable": (C.c_int, [C.c_int]),
    "hn_range_check_fetch": (C.c_int, [c_i32p, C.c_int, VP]),
    "hn_range_check_enabled": (C.c_int, []),
    "hn_range_check_bind": (C.c_int, [VP]),
    "hn_range_scope_begin": (C.c_int, [VP, C.c_int]),
    "hn_range_scope_end": (C.c_int, []),
    "hn_range_check_collect": (C.c_int, [VP, VP, VP]),
    "hn_set_form": (C.c_int, [C.c_char_p, C.c_int]),
    "hn_set_tuning": (C.c_int, [C.c_char_p, C.c_double]),
    "hn_conv2d_nhwc_f32": (C.c_int, [C.POINTER(ConvDesc), VP, VP, VP, VP, VP, VP, VP, VP]),
    "hn_conv2d_pick_tile": (C.c_int, [C.POINTER(ConvDesc)]),
    "hn_conv2d_nhwc_f16x3": (C.c_int, [C.POINTER(ConvDesc), VP, VP, VP, VP, VP, VP]),
    "hn_affine_split_f32": (C.c_int, [VP, VP, VP] + [C.c_int] * 6 + [VP, C.c_int, VP]),
    "hn_unsplit_f32": (C.c_int, [VP] + [C.c_int] * 4 + [VP, C.c_int, VP]),
    "hn_maxpool3x3s2_s32": (C.c_int, [VP, VP] + [C.c_int] * 6 + [VP]),
    "hn_conv2d_f16x3_pick_tile": (C.c_int, [C.POINTER(ConvDesc)]),
    "hn_conv2d_f16x3_uses_stream": (C.c_int, [C.POINTER(ConvDesc)]),
    "hn_conv2d_f16x3_uses_rs": (C.c_int, [C.POINTER(ConvDesc)]),
    "hn_conv2d_f16x3_uses_halo": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "hn_maxpool3x3s2_nhwc_f32": (C.c_int, [VP, VP] + [C.c_int] * 6 + [VP]),
    "hn_groupnorm_scratch_floats": (C.c_int64, [C.c_int] * 4),
    "hn_groupnorm_affine_f32": (C.c_int, [VP, VP, VP] + [C.c_int] * 4 + [C.c_float, VP, VP, VP, VP]),
    "hn_fcos_preprocess_f32": (C.c_int, [VP, VP] + [C.c_int] * 7 + [c_f32p, c_f32p, VP]),
    "hn_conv2d_nhwc_f16x3_ws": (C.c_int, [C.POINTER(ConvDesc)] + [VP] * 6 + [C.c_int64, VP]),
    "hn_conv2d_nhwc_f16x3_grouped": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvGroup), VP]),
    "hn_conv2d_nhwc_f16x3_gn": (C.c_int, [C.POINTER(ConvDesc)] + [VP] * 6),
    "hn_groupnorm_rows32_scratch_floats": (C.c_int64, [C.c_int64, C.c_int]),
    "hn_groupnorm_finalize_rows32": (C.c_int, [VP, VP, VP] + [C.c_int] * 4 + [C.c_float, VP, VP, VP]),
    "hn_spmm_csr_f32": (C.c_int, [VP, VP, VP, C.c_int, VP, VP, C.c_int, C.c_int, VP]),
    "hn_graph_conv_cheby3_f16x3": (C.c_int, [C.POINTER(GraphCsr), C.POINTER(GraphCsr), VP, C.c_int, C.c_int, VP, C.c_int, VP,
                                             C.c_int, C.c_int, VP, C.c_int, C.c_int, VP, C.c_int, VP]),
    "hn_linear_rows_f16x3": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, VP,
                                       C.c_int, VP]),
    "hn_mesh_finish_f32": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "hn_pad_split_rows_f32": (C.c_int, [VP, C.c_int64, C.c_int, C.c_int, VP, VP]),
    "hn_lifter_combine_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "hn_cheby3_basis_split": (C.c_int, [VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "hn_feat_interp_add_f32": (C.c_int, [VP, VP, VP, C.c_int64, C.c_int, C.c_int, C.c_int, VP]),
    "hn_fcos_preprocess_split": (C.c_int, [VP, VP] + [C.c_int] * 8 + [c_f32p, c_f32p, VP]),
    "hn_fcos_preprocess_list": (C.c_int, [VP, VP, VP] + [C.c_int] * 5 + [c_f32p, c_f32p, VP]),
    "hn_conv_stem_f16x3": (C.c_int, [VP] + [C.c_int] * 7 + [VP, VP, C.c_int, VP, C.c_int, VP]),
    "hn_conv_stem_pool_f16x3": (C.c_int, [VP] + [C.c_int] * 7 + [VP, VP, VP, VP]),
    "hn_conv_stem_pool_f16x3_terms": (C.c_int, [VP] + [C.c_int] * 7 + [VP, VP, VP, C.c_int, VP]),
    "hn_fcos_candidates": (C.c_int, [C.POINTER(FcosLevels), C.c_int, C.c_int, C.c_float,
                                     VP, VP, VP, VP, VP, VP, VP, C.c_int, VP]),
    "hn_fcos_candidates_ws_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "hn_fcos_candidates_ws": (C.c_int, [C.POINTER(FcosLevels), C.c_int, C.c_int, C.c_float,
                                     VP, VP, VP, VP, VP, VP, VP, C.c_int, VP, C.c_int64, VP]),
    "hn_fcos_ext_gather": (C.c_int, [C.POINTER(FcosLevels), C.POINTER(VP), VP, VP, VP, C.c_int, C.c_int, VP, VP, VP]),
    "hn_fcos_nms_scratch_bytes": (C.c_int64, [C.c_int, C.c_int]),
    "hn_fcos_nms": (C.c_int, [VP] * 6 + [C.c_int, C.c_int, C.c_double, C.c_float, C.c_float] + [VP] * 9),
    "hn_fcos_nms_ratios": (C.c_int, [VP] * 6 + [C.c_int, C.c_int, C.c_double, VP] + [VP] * 9),
    "hn_nms": (C.c_int, [VP, VP, C.c_int, C.c_double, VP, VP, VP, VP]),
    "hn_crop_resize": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, VP] + [C.c_int] * 7 + [VP, VP, VP, VP]),
    "hn_stem_image_nhwc4": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "hn_stem_image_nhwc4_valid": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP]),
    "hn_pack_depth_nhwc": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, VP]),
    "hn_a2j_aggregate_f32": (C.c_int, [VP, VP, VP, VP] + [C.c_int] * 5 + [VP, VP]),
    "hn_create": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(VP)]),
    "hn_load_weight": (C.c_int, [VP, C.c_char_p, VP, c_i64p, C.c_int]),
    "hn_finalize": (C.c_int, [VP]),
    "hn_fcos_capacity": (C.c_int64, [VP, C.c_int, C.c_int]),
    "hn_fcos_forward": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int] + [VP] * 6 + [C.c_int, VP]),
    "hn_fcos_forward_ext": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int] + [VP] * 8 + [C.c_int, VP]),
    "hn_fcos_capacity_list": (C.c_int64, [VP, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "hn_fcos_forward_list": (C.c_int, [VP, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int,
                                       VP, VP, VP, VP, VP, VP, C.c_int, VP]),
    "hn_a2j_forward": (C.c_int, [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, VP]),
    "hn_handnet_forward": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP]),
    "hn_destroy": (C.c_int, [VP]),
    "hn_groupnorm_finalize_rows32_levels": (C.c_int, [C.POINTER(GnLevels), VP, VP, C.c_int, C.c_int, C.c_int, C.c_float, VP]),
    "hn_affine_split_f32_levels": (C.c_int, [C.POINTER(SplitLevels)] + [C.c_int] * 6 + [VP]),
    "hn_pack_records": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP]),
    "hn_debug_tickets_nonzero": (C.c_int, [c_i64p]),
    "hn_pack_records_ex": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP]),
    "hn_unpack_records": (C.c_int, [VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, VP, VP]),
    "hn_nonfinite_count_f32": (C.c_int, [VP, C.c_int64, VP, VP]),
    "hn_conv3x3_thin_f16x3_levels": (C.c_int, [C.POINTER(ThinLevels), C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, C.c_int, VP]),
    "hn_conv3x3_thin_f16x3_levels_group": (C.c_int, [C.POINTER(ThinMember), C.c_int, C.c_int, C.c_int, C.c_int, VP]),
    "hn_conv3x3_thin_affine_applies": (C.c_int, [C.POINTER(ThinLevels), C.c_int, C.c_int, C.c_int]),
    "hn_conv3x3_thin_affine_f16x3_levels": (C.c_int, [C.POINTER(ThinLevels), C.POINTER(ThinAffine), C.c_int, C.c_int, C.c_int,
                                                      VP, VP, C.c_int, VP]),
    "hn_conv3x3_thin_uses_flat": (C.c_int, [C.POINTER(ThinLevels), C.c_int, C.c_int, C.c_int]),
    "hn_convert_joints_f32": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_float, C.c_float, c_f32p, VP, VP]),
    "hn_convert_joints_samples_f32": (C.c_int, [VP, VP, VP, VP, C.c_int, C.c_int, C.c_float, C.c_float, VP, VP, VP]),
    "hn_a2j_aggregate_convert_f32": (C.c_int, [VP, VP, VP, VP] + [C.c_int] * 5 + [VP, C.c_float, C.c_float, c_f32p,
                                               C.POINTER(ConvertOpts), VP, VP, VP, VP]),
    "hn_joints2d_standardize_f32": (C.c_int, [VP, VP, C.c_int, C.c_int, VP, VP]),
    "hn_handnet_forward_xyz": (C.c_int, [VP, VP, VP, C.c_int, C.c_int, C.c_int, c_f32p, C.POINTER(ConvertOpts),
                                         VP, VP, VP, VP, VP, VP]),
}

_lock = threading.Lock()
_lib = None


def lib_path() -> Path:
    """The in-tree library; HN_LIB_PATH selects another build (ablation / probe variants under tools/) without
    touching the product file."""
    override = os.environ.get("HN_LIB_PATH")
    return Path(override) if override else _build.LIB_PATH


def load(build_if_missing: bool = True) -> C.CDLL:
    """Load (building first if needed) the HIP library.  Raises on any failure."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = lib_path()
        if not path.exists():
            if not build_if_missing or path != _build.LIB_PATH:
                raise RuntimeError(f"{path} is missing; run `python __graft_entry__.py build`")
            _build.build_library()
        lib = C.CDLL(str(path))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise RuntimeError(f"{path} does not export {name}; rebuild it") from e
            fn.restype, fn.argtypes = res, args
        if lib.hn_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI {lib.hn_abi_version()} != expected {ABI_VERSION}; rebuild it")
        _lib = lib
        return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().hn_last_error()
        raise RuntimeError(f"{what} failed (status {status}): {msg.decode() if msg else '?'}")


def ptr(t) -> int | None:
    """Device/host address of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
