"""ctypes binding of librsgnet_hip.so (the C ABI in include/rsgnet_hip.h).

Loading fails loudly: there is no fallback implementation of any entry point.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

RSG_F16, RSG_BF16 = 0, 1
RSG_LAYOUT_OUT_PLANES, RSG_LAYOUT_RES_PLANES, RSG_LAYOUT_IN_PLANES = 1, 2, 4      # rsg_conv_desc.layout
DTYPES = {"fp16": RSG_F16, "bf16": RSG_BF16}


class ConvDesc(C.Structure):
    """struct rsg_conv_desc (include/rsgnet_hip.h)."""
    _fields_ = [
        ("inp", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
        ("out16", C.c_void_p), ("out_nchw_f32", C.c_void_p),
        ("batch", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("in_c", C.c_int32), ("in_ld", C.c_int32),
        ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_c", C.c_int32), ("out_ld", C.c_int32), ("res_ld", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad_y", C.c_int32), ("pad_x", C.c_int32),
        ("up", C.c_int32), ("up_y", C.c_int32), ("up_x", C.c_int32), ("phys_h", C.c_int32), ("phys_w", C.c_int32),
        ("relu", C.c_int32), ("dtype", C.c_int32), ("tile_hint", C.c_int32), ("relu_tile_mask", C.c_int32),
        ("out_c_store", C.c_int32), ("layout", C.c_int32),
        ("residual2", C.c_void_p), ("residual3", C.c_void_p), ("res2_ld", C.c_int32), ("res3_ld", C.c_int32),
    ]


class FuseItem(C.Structure):
    """struct rsg_fuse_item (include/rsgnet_hip.h)."""
    _fields_ = [("same", C.c_void_p * 4), ("low", C.c_void_p * 3), ("low_shift", C.c_int32 * 3),
                ("n_same", C.c_int32), ("n_low", C.c_int32), ("out", C.c_void_p), ("batch", C.c_int32),
                ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32), ("same_ld", C.c_int32 * 4),
                ("low_ld", C.c_int32 * 3), ("reserved", C.c_int32)]


class FmConv(C.Structure):
    """struct rsg_fm_conv (include/rsgnet_hip.h)."""
    _fields_ = [("w_packed", C.c_void_p), ("bias", C.c_void_p), ("out16", C.c_void_p), ("out_ld", C.c_int32),
                ("src", C.c_int32), ("k", C.c_int32), ("cout", C.c_int32), ("relu", C.c_int32), ("role", C.c_int32),
                ("reserved", C.c_int32)]


FM_MAX_CONVS = 12
FM_TO_GLOBAL, FM_UP_Y0, FM_UP_Y1, FM_DOWN_Y1 = 0, 1, 2, 3


class FuseModuleDesc(C.Structure):
    """struct rsg_fuse_module_desc (include/rsgnet_hip.h)."""
    _fields_ = [("x", C.c_void_p * 4), ("y", C.c_void_p * 2), ("x_ld", C.c_int32 * 4), ("y_ld", C.c_int32 * 2),
                ("batch", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("n_branches", C.c_int32), ("dtype", C.c_int32),
                ("y1_fold", C.c_int32), ("n_convs", C.c_int32), ("c0", C.c_int32), ("convs", FmConv * FM_MAX_CONVS)]


class ChainItem(C.Structure):
    """struct rsg_chain_item (include/rsgnet_hip.h)."""
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("w_packed", C.POINTER(C.c_void_p)),
                ("bias", C.POINTER(C.c_void_p)), ("nlayers", C.c_int32), ("batch", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("c", C.c_int32), ("images_per_workgroup", C.c_int32), ("workspace", C.c_void_p),
                ("done_flags", C.c_void_p), ("wait_flags", C.c_void_p), ("split_rows", C.c_int32), ("exchange", C.c_int32),
                ("in_ld", C.c_int32), ("out_ld", C.c_int32)]


class PackItem(C.Structure):
    """struct rsg_pack_item (include/rsgnet_hip.h)."""
    _fields_ = [("w", C.c_void_p), ("dst", C.c_void_p), ("cin", C.c_int32), ("cout", C.c_int32), ("kh", C.c_int32),
                ("kw", C.c_int32), ("transpose", C.c_int32), ("first_block", C.c_int32)]


class BnItem(C.Structure):
    """struct rsg_bn_item (include/rsgnet_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("y", "dz", "z", "residual", "out", "dres", "gamma", "beta", "mean", "invstd",
                                          "running_mean", "running_var", "dgamma", "dbeta", "workspace")] + \
               [("pixels", C.c_int64), ("c", C.c_int32), ("relu", C.c_int32)]


class WgradItem(C.Structure):
    """struct rsg_wgrad_item (include/rsgnet_hip.h)."""
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("workspace", C.c_void_p)] + \
               [(n, C.c_int32) for n in "batch in_h in_w cin cin_pad out_h out_w cout cout_pad kh kw stride pad reserved".split()]


class CropDesc(C.Structure):
    """struct rsg_crop_desc (include/rsgnet_hip.h)."""
    _fields_ = [("image", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("stride", C.c_int32),
                ("reserved", C.c_int32), ("m", C.c_double * 6)]


# every symbol include/rsgnet_hip.h declares: name -> (restype, argtypes)
_vp, _i, _fp = C.c_void_p, C.c_int32, C.c_void_p
SYMBOLS = {
    "rsg_last_error": (C.c_char_p, []),
    "rsg_version": (C.c_int, []),
    "rsg_tune": (C.c_int, [C.c_char_p, _i]),
    "rsg_conv_igemm": (C.c_int, [C.POINTER(ConvDesc), _vp]),
    "rsg_stem_fused": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_stem_fused_conv1": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_conv_dual": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _i, _i, _vp]),
    "rsg_conv_pair": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _i, _vp, _i, _vp, _i, _i, _vp]),
    "rsg_conv_pair_next_perm": (C.c_int32, [_i]),
    "rsg_conv_tail1x1": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _i, _vp, _vp]),
    "rsg_conv_pair_next": (C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "rsg_bottleneck_head_perm": (C.c_int32, [_i]),
    "rsg_bottleneck_fused": (C.c_int, [C.POINTER(ConvDesc), _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "rsg_conv_plan": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int32)]),
    "rsg_conv_padded_cin": (C.c_int32, [_i]),
    "rsg_conv_packed_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "rsg_pack_conv_weights_host": (C.c_int, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsg_graph_begin": (C.c_int, [C.POINTER(_vp)]),
    "rsg_graph_next_deps": (C.c_int, [_vp, C.POINTER(C.c_int32), _i]),
    "rsg_graph_node_count": (C.c_int, [_vp]),
    "rsg_graph_end": (C.c_int, [_vp]),
    "rsg_graph_abort": (C.c_int, [_vp]),
    "rsg_graph_launch": (C.c_int, [_vp, _vp]),
    "rsg_graph_destroy": (C.c_int, [_vp]),
    "rsg_basic_chain": (C.c_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsg_basic_chain_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i, _i]),
    "rsg_basic_chain_group": (C.c_int, [C.POINTER(ChainItem), _i, _i, _vp]),
    "rsg_basic_chain_exchange_bytes": (C.c_size_t, [_i, _i, _i, _i, _i, C.POINTER(C.c_int32)]),
    "rsg_chain_handoff_timeouts": (C.c_int, [C.POINTER(C.c_uint32)]),
    "rsg_chain_handoff_timeouts_async": (C.c_int, [_vp, _vp]),
    "rsg_basic_chain_handoff_max_waiters": (C.c_int, []),
    "rsg_basic_chain_supported": (C.c_int, [_i, _i, _i, _i, _i, C.POINTER(C.c_int32)]),
    "rsg_basic_chain_supported_split": (C.c_int, [_i, _i, _i, _i, _i, C.POINTER(C.c_int32)]),
    "rsg_conv_groupable": (C.c_int, [C.POINTER(ConvDesc)]),
    "rsg_sgm_graph_conv": (C.c_int, [_vp] * 10 + [_i, _i, _i, C.c_float, _vp]),
    "rsg_sgm_graph_conv_backward": (C.c_int, [_vp] * 14 + [_i, _i, _i, C.c_float, _vp]),
    "rsg_fuse_module_table_bytes": (C.c_size_t, []),
    "rsg_fuse_module_plan": (C.c_int, [C.POINTER(FuseModuleDesc), _vp, C.POINTER(C.c_int32)]),
    "rsg_fuse_module_run": (C.c_int, [_vp, _i, _i, _i, _vp]),
    "rsg_fuse_module_table_jobs": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _vp]),
    "rsg_group_table_bytes": (C.c_size_t, []),
    "rsg_conv_fuse_group_plan": (C.c_int, [C.POINTER(ConvDesc), _i, C.POINTER(FuseItem), _i, _i, _vp,
                                           C.POINTER(C.c_int32)]),
    "rsg_conv_fuse_group_run": (C.c_int, [_vp, _i, _i, _i, _vp]),
    "rsg_nchw_f32_to_nhwc16": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsg_nchw_f32_to_s2d_nhwc16": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_bilinear2x_backward_nchw_f32": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "rsg_trp_tail_backward_partial_floats": (C.c_size_t, [_i, _i, _i]),
    "rsg_trp_tail_backward": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_trp_backward": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_maxpool2x2_nhwc16": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsg_f32_rows_to_nhwc16": (C.c_int, [_vp, _vp, _i, C.c_int64, _i, _i, _vp]),
    "rsg_trp_tail_in16": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsg_hr_fuse_sum": (C.c_int, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(C.c_int32), _i, _vp,
                                  _i, _i, _i, _i, _i, _vp]),
    "rsg_fuse_classifier": (C.c_int, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_fuse_head4": (C.c_int, [_vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_flip_merge_nchw_f32": (C.c_int, [_vp, _vp, C.POINTER(C.c_int32), _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_decode_heatmaps": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "rsg_bilinear2x_nchw_f32": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_bilinear2x_pair_nchw_f32": (C.c_int, [_vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_trp_attention": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_trp_project_g": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_trp_tail_workspace_bytes": (C.c_size_t, [_i, _i, _i]),
    "rsg_trp_tail": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsg_trp_tail_ex": (C.c_int, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "rsg_trp_scores": (C.c_int, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "rsg_loss_workspace_bytes": (C.c_size_t, [_i]),
    "rsg_joints_mse_loss": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, C.c_float, _vp, _i, _vp, _vp, _vp]),
    "rsg_bce_loss": (C.c_int, [_vp, _vp, C.c_int64, C.c_float, C.c_float, _vp, _i, _vp, _vp, _vp]),
    "rsg_relation_target_vec": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_adam_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _i, C.c_float, _vp]),
    "rsg_pack_item_blocks": (C.c_int32, [_i, _i, _i, _i]),
    "rsg_pack_conv_weights_device": (C.c_int, [_vp, _i, _i, _i, _vp]),
    "rsg_bn_workspace_bytes": (C.c_size_t, [_i]),
    "rsg_bn_stats": (C.c_int, [_vp, C.c_int64, _i, _i, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp]),
    "rsg_bn_apply": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, C.c_int64, _i, _i, _vp]),
    "rsg_bn_backward": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _i, _i, _vp]),
    "rsg_bn_forward_group": (C.c_int, [C.POINTER(BnItem), _i, C.c_float, C.c_float, _i, _vp]),
    "rsg_bn_backward_group": (C.c_int, [C.POINTER(BnItem), _i, _i, _vp]),
    "rsg_relu_backward": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i, _vp]),
    "rsg_add16": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i, _vp]),
    "rsg_upsample_nearest_backward": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "rsg_zero_insert2x": (C.c_int, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_channel_sum_workspace_bytes": (C.c_size_t, [_i]),
    "rsg_channel_sum_nchw_f32": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "rsg_conv_wgrad_workspace_bytes": (C.c_size_t, [_i] * 11),
    "rsg_conv_wgrad": (C.c_int, [_vp, _vp, _vp, _vp] + [_i] * 14 + [_vp]),
    "rsg_conv_wgrad_group": (C.c_int, [C.POINTER(WgradItem), _i, _i, _vp]),
    "rsg_conv_wgrad_group_ex": (C.c_int, [C.POINTER(WgradItem), _i, C.POINTER(WgradItem), _i, _i, _i, _vp]),
    "rsg_wgrad_set_quad": (None, [_i]),
    "rsg_chain48_packed_bytes": (C.c_size_t, []),
    "rsg_pack_chain48_weights_host": (C.c_int, [_vp, _vp, _vp, _i, _vp, _vp]),
    "rsg_chain_pack_kind": (C.c_int, [_i]),
    "rsg_pack_chain16_weights_host": (C.c_int, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "rsg_wgrad_tune": (None, [_i, _i, _i, _i, _i]),
    "rsg_crop_warp_affine": (C.c_int, [_vp, _vp, _vp, _i, _i, _i, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _vp]),
    "rsg_trp_mse_workspace_bytes": (C.c_size_t, [_i, _i]),
    "rsg_trp_relation_mse": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_trp_relation_mse_dense": (C.c_int, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_trp_backward_dense": (C.c_int, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "rsg_rows_project_f32": (C.c_int, [_vp, _vp, _vp, C.c_int64, _i, _i, _vp]),
    "rsg_rows_gram_workspace_bytes": (C.c_size_t, [_i]),
    "rsg_rows_gram_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, _i, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None
# include/rsgnet_hip.h, RSG_ABI_VERSION: bumped with every incompatible change of an entry point's arguments or of a struct
# (2 = round 5: rsg_trp_relation_mse's workspace argument and the slimmer rsg_chain_item of round 4, rsg_fuse_module)
ABI_VERSION = 2


def lib_path() -> str:
    """The in-tree library; RSGNET_HIP_LIB names another build of the same sources (A/B runs)."""
    return os.environ.get("RSGNET_HIP_LIB") or _build.LIB


def lib() -> C.CDLL:
    """Load (once) the HIP library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -m rsgnet_amd.build` (hipcc, gfx950). "
                "rsgnet_amd has no non-HIP fallback.")
        # the in-tree library must be the build of the sources in the tree (a failed rebuild leaves the previous .so behind,
        # and a measurement or a green test of THAT library says nothing about the sources); a library named through
        # RSGNET_HIP_LIB is another build on purpose (A/B runs)
        # (name="": the library that is CHECKED is the one that is LOADED, whatever RSGNET_HIP_BUILD_NAME says -- that
        # variable names the variant `python -m rsgnet_amd.build` writes, for RSGNET_HIP_LIB to pick up)
        if not os.environ.get("RSGNET_HIP_LIB") and os.environ.get("RSGNET_HIP_ALLOW_STALE") != "1" and _build.stale(name=""):
            if os.environ.get("RSGNET_HIP_NO_AUTOBUILD") == "1":
                raise RuntimeError(
                    f"{path} was not built from the sources, flags and compiler in this tree ({_build.describe('')}): "
                    "run `python -m rsgnet_amd.build` (RSGNET_HIP_ALLOW_STALE=1 loads it anyway)")
            # rebuilds (about 90 s) or raises hipcc's error: never the previous build in silence.  Ranks / workers that get
            # here together are serialised by build()'s lock file; all but the first find a fresh library behind it.
            _build.build(name="")
        l = C.CDLL(path)
        l.rsg_version.restype = C.c_int
        if l.rsg_version() != ABI_VERSION:
            raise RuntimeError(f"{path} speaks C-ABI version {l.rsg_version()}, this package binds version {ABI_VERSION} "
                               "(include/rsgnet_hip.h): rebuild with `python -m rsgnet_amd.build --force`")
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)      # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        # RSGNET_HIP_TUNE="key=value,...": the library's knobs (rsg_tune validates the values; the wrong-result timing switches
        # chain_debug bits 1|2 and bottleneck_debug exist only in a -DRSG_DEBUG build, RSGNET_HIP_DEBUG_BUILD=1), and the
        # weight-gradient tiling (rsg_wgrad_tune / rsg_wgrad_set_quad) under the wgrad_ prefix
        from . import switches as _sw
        tune = _sw.tune_items()
        wg = {k[6:]: v for k, v in tune.items() if k.startswith("wgrad_")}
        for key, value in tune.items():
            if key.startswith("wgrad_"):
                continue
            if l.rsg_tune(key.encode(), value):
                raise RuntimeError(f"RSGNET_HIP_TUNE {key}={value}: {l.rsg_last_error().decode(errors='replace')}")
        if any(k in wg for k in ("min_px", "quad_min_wgs", "lds", "px_cap", "waves")):
            l.rsg_wgrad_tune(wg.get("min_px", 0), wg.get("quad_min_wgs", 0), wg.get("lds", 0), wg.get("px_cap", 0), wg.get("waves", 0))
        if "quad" in wg:
            l.rsg_wgrad_set_quad(wg["quad"])
        _lib = l
    return _lib


def pack_chain_layer(w, scale, shift, dtype: int):
    """One 3x3 C->C layer of a BasicBlock chain in the order the chain kernels of its width read
    (rsg_chain_pack_kind: 16x16x32 pack for HRNet-W32's widths, the 48-channel pack, else the conv pack).
    w fp32 [C,C,3,3], scale / shift fp32 [C] or None (folded BatchNorm); returns (weights u8, bias f32) as numpy arrays."""
    import numpy as np
    l = lib()
    w = np.ascontiguousarray(w, dtype=np.float32)
    c = w.shape[0]
    if w.shape != (c, c, 3, 3):
        raise ValueError(f"pack_chain_layer: weight of shape {w.shape}")
    sc = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    sh = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    ps, ph = (sc.ctypes.data if sc is not None else None), (sh.ctypes.data if sh is not None else None)
    kind = l.rsg_chain_pack_kind(c)
    if kind == 16:
        wp, bo = np.empty(18 * c * c, dtype=np.uint8), np.empty(c, dtype=np.float32)
        check(l.rsg_pack_chain16_weights_host(w.ctypes.data, ps, ph, c, dtype, wp.ctypes.data, bo.ctypes.data), "pack_chain16")
    elif kind == 48:
        wp, bo = np.empty(l.rsg_chain48_packed_bytes(), dtype=np.uint8), np.empty(48, dtype=np.float32)
        check(l.rsg_pack_chain48_weights_host(w.ctypes.data, ps, ph, dtype, wp.ctypes.data, bo.ctypes.data), "pack_chain48")
    else:
        wp = np.empty(l.rsg_conv_packed_bytes(c, c, 3, 3), dtype=np.uint8)
        bo = np.empty((c + 31) // 32 * 32, dtype=np.float32)
        check(l.rsg_pack_conv_weights_host(w.ctypes.data, ps, ph, None, c, c, 3, 3, dtype, wp.ctypes.data, bo.ctypes.data), "pack")
    return wp, bo


def check(rc: int, what: str = "") -> None:
    if rc:
        msg = lib().rsg_last_error().decode(errors="replace")
        raise RuntimeError(f"librsgnet_hip {what}: {msg}")
