"""ctypes binding of libsegmi.so (C ABI declared in include/segmi.h).

The library is loaded AFTER torch so that its DT_NEEDED libamdhip64.so.7 resolves to the HIP runtime
torch already mapped (one runtime per process: torch's streams / device pointers stay valid for our
kernels).  There is no fallback: if the shared object is missing or a symbol is absent, importing
this module raises.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be imported first, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsegmi.so")

vp, i32, i64, f32, u64, sz = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64, C.c_size_t


class ConvDesc(C.Structure):
    """struct segmi_conv_desc"""
    _fields_ = [(n, i32) for n in ("N", "H", "W", "C", "K", "R", "S", "P", "Q", "stride", "pad", "dil", "ldx", "ldy")]


PD = C.POINTER(ConvDesc)


class FilterTx(C.Structure):
    """struct segmi_filter_tx (one entry of the batched filter transposition)"""
    _fields_ = [("w_krsc", vp), ("w_crsk", vp), ("K", i32), ("R", i32), ("S", i32), ("C", i32), ("Kpad", i32), ("tile_begin", i32)]


# name -> (restype, argtypes); mirrors include/segmi.h one to one (tests/test_abi.py checks it)
SIGNATURES = {
    "segmi_strerror": (C.c_char_p, [i32]),
    "segmi_abi_version": (i32, []),
    "segmi_nchw_to_nhwc": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "segmi_nhwc_to_nchw": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "segmi_copy_rows": (i32, [vp, i32, vp, i32, i64, i32, i32, vp]),
    "segmi_conv2d_fwd_workspace": (sz, [PD]),
    "segmi_conv2d_fwd": (i32, [PD, vp, vp, vp, vp, i32, vp, sz, vp]),
    "segmi_conv2d_fwd_stats_parts": (i32, [PD]),
    "segmi_conv2d_fwd_stats": (i32, [PD, vp, vp, vp, vp, vp, vp]),
    "segmi_conv2d_dgrad": (i32, [PD, vp, vp, vp, i32, vp]),
    "segmi_conv2d_wgrad_workspace": (sz, [PD]),
    "segmi_conv2d_wgrad": (i32, [PD, vp, vp, vp, vp, sz, vp]),
    "segmi_conv2d_variant": (i32, [PD, i32, C.c_char_p, sz]),
    "segmi_filter_krsc_to_crsk": (i32, [vp, vp, i32, i32, i32, i32, i32, vp]),
    "segmi_conv2d_winograd_ok": (i32, [PD, i32]),
    "segmi_conv2d_winograd_workspace": (sz, [PD, i32]),
    "segmi_conv2d_winograd_fwd": (i32, [PD, vp, vp, vp, vp, i32, vp, vp, vp, sz, vp]),
    "segmi_conv2d_winograd_fwd_stats_parts": (i32, [PD]),
    "segmi_conv2d_winograd_v_bytes": (sz, [PD]),
    "segmi_conv2d_winograd_dgrad": (i32, [PD, vp, vp, vp, i32, vp, sz, vp]),
    "segmi_conv2d_winograd_variant": (i32, [PD, i32, C.c_char_p, sz]),
    "segmi_conv2d_winograd_wgrad_ok": (i32, [PD]),
    "segmi_conv2d_winograd_wgrad_workspace": (sz, [PD]),
    "segmi_conv2d_winograd_wgrad": (i32, [PD, vp, vp, vp, vp, vp, sz, vp]),
    "segmi_conv2d_winograd_wgrad_variant": (i32, [PD, C.c_char_p, sz]),
    "segmi_conv2d_winograd_tiles": (i64, [PD]),
    "segmi_conv2d_winograd_trace": (i32, [vp, vp]),
    "segmi_filter_tx_tiles": (i64, [i32, i32, i32, i32, i32]),
    "segmi_filter_krsc_to_crsk_multi": (i32, [vp, i32, i64, vp]),
    "segmi_dwconv2d_fwd": (i32, [PD, vp, vp, vp, vp]),
    "segmi_dwconv2d_fwd_stats_parts": (i32, [PD]),
    "segmi_dwconv2d_fwd_stats": (i32, [PD, vp, vp, vp, vp, vp]),
    "segmi_dwconv2d_dgrad": (i32, [PD, vp, vp, vp, vp]),
    "segmi_dwconv2d_pre_ok": (i32, [PD]),
    "segmi_dwconv2d_fwd_pre": (i32, [PD, vp, vp, vp, i32, vp, vp, vp, vp]),
    "segmi_dwconv2d_wgrad_pre": (i32, [PD, vp, vp, vp, i32, vp, vp, vp, sz, vp]),
    "segmi_dwconv2d_wgrad_workspace": (sz, [PD]),
    "segmi_dwconv2d_wgrad": (i32, [PD, vp, vp, vp, vp, sz, vp]),
    "segmi_depth_to_space2": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, vp]),
    "segmi_space_to_depth2": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "segmi_colsum_workspace": (sz, [i64, i32]),
    "segmi_colsum": (i32, [vp, i32, i64, i32, vp, vp, sz, vp]),
    "segmi_bn_stats_workspace": (sz, [i64, i32]),
    "segmi_bn_stats": (i32, [vp, i32, i64, i32, vp, vp, sz, vp]),
    "segmi_bn_finalize": (i32, [vp, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "segmi_bn_stats_finalize": (i32, [vp, i32, i64, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "segmi_bn_parts_workspace": (sz, [i32, i32]),
    "segmi_bn_stats_from_parts": (i32, [vp, i32, i32, vp, vp, sz, vp]),
    "segmi_bn_finalize_from_parts": (i32, [vp, i32, i32, vp, vp, f32, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]),
    "segmi_bn_eval_coeffs": (i32, [vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp]),
    "segmi_bn_apply": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, vp, vp, i32, vp]),
    "segmi_bn_bwd_reduce_workspace": (sz, [i64, i32]),
    "segmi_bn_bwd_reduce": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp, vp, sz, vp]),
    "segmi_bn_bwd_apply": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, vp, vp, vp, vp, vp, f32, vp, i32, i32, vp, i32, vp, i32, vp]),
    "segmi_relu_fwd": (i32, [vp, i32, vp, i32, i64, i32, vp]),
    "segmi_relu_bwd": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, vp]),
    "segmi_add": (i32, [vp, i32, vp, i32, vp, i32, i64, i32, vp]),
    "segmi_maxpool_fwd": (i32, [vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "segmi_maxpool_bwd": (i32, [vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "segmi_adaptive_avgpool_fwd": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "segmi_adaptive_avgpool_bwd": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "segmi_bilinear_fwd": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "segmi_bilinear_bwd_workspace": (sz, [i32, i32, i32, i32, i32, i32]),
    "segmi_bilinear_bwd": (i32, [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, sz, vp]),
    "segmi_dropout": (i32, [vp, i32, vp, i32, i32, i64, i32, f32, i32, u64, vp, vp]),
    "segmi_filter_slice": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "segmi_filter_unslice": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "segmi_pyramid_up_workspace": (sz, [i32, i32, i32, i32, i32, vp, vp]),
    "segmi_pyramid_up_fwd": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp, sz, vp]),
    "segmi_pyramid_up_bwd": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "segmi_comm_available": (i32, []),
    "segmi_comm_unique_id_bytes": (i32, []),
    "segmi_comm_get_unique_id": (i32, [vp, sz]),
    "segmi_comm_init": (i32, [C.POINTER(vp), i32, i32, vp, sz]),
    "segmi_comm_world": (i32, [vp]),
    "segmi_comm_allreduce_async": (i32, [vp, vp, vp, sz, i32, C.POINTER(C.c_long), vp]),
    "segmi_comm_allgather_async": (i32, [vp, vp, vp, sz, C.POINTER(C.c_long), vp]),
    "segmi_comm_wait_ticket": (i32, [vp, C.c_long, vp]),
    "segmi_comm_wait": (i32, [vp, vp]),
    "segmi_comm_destroy": (i32, [vp]),
    "segmi_aug_resize": (i32, [vp, vp, i32, i32, vp, vp, i32, i32, vp, i32, vp]),
    "segmi_aug_rotate": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
    "segmi_aug_blur": (i32, [vp, i32, i32, i32, i32, vp, vp, vp]),
    "segmi_aug_finish": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp, i32, vp, vp]),
    "segmi_ce_workspace": (sz, [i64]),
    "segmi_ce_fwd": (i32, [vp, i32, vp, i64, i32, i64, vp, vp, vp, vp, sz, vp]),
    "segmi_ce_bwd": (i32, [vp, i32, vp, vp, i64, i32, i64, vp, vp, vp, vp, i32, vp]),
    "segmi_upsample_ce_workspace": (sz, [i32, i32, i32, i32, i32, i32]),
    "segmi_upsample_ce_fwd": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp, vp, vp, vp, sz, vp]),
    "segmi_upsample_ce_bwd": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, i64, vp, vp, vp, vp, i32, vp, sz, vp]),
    "segmi_dice_workspace": (sz, [i64]),
    "segmi_dice_fwd": (i32, [vp, i32, vp, i64, i32, i64, f32, vp, vp, vp, vp, sz, vp]),
    "segmi_dice_bwd": (i32, [vp, i32, vp, vp, i64, i32, vp, vp, vp, i32, vp]),
    "segmi_target_stats": (i32, [vp, i64, i64, vp, vp, sz, vp]),
    "segmi_dice_sums": (i32, [vp, i32, vp, i64, i32, i64, vp, vp, vp, vp, sz, vp]),
    "segmi_dice_finalize": (i32, [vp, f32, vp, vp]),
    "segmi_focal_fwd": (i32, [vp, i32, vp, i64, i32, i64, f32, vp, vp, vp, vp, sz, vp]),
    "segmi_focal_bwd": (i32, [vp, i32, vp, vp, i64, i32, i64, f32, vp, vp, vp, vp, i32, vp]),
    "segmi_lovasz_set_prune": (i32, [i32]),
    "segmi_lovasz_workspace": (sz, [i64, i32]),
    "segmi_lovasz_fwd": (i32, [vp, i32, vp, i64, i32, i64, vp, vp, i32, vp, vp, sz, vp]),
    "segmi_lovasz_bwd": (i32, [vp, i32, vp, i64, vp, vp, i32, i64, i32, vp, vp, vp, i32, vp]),
    "segmi_upsample_lovasz_workspace": (sz, [i32, i32, i32, i32, i32, i32]),
    "segmi_upsample_lovasz_fwd": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp, vp, i32, vp, vp, sz, vp]),
    "segmi_upsample_lovasz_bwd": (i32, [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, i64, vp, vp, i32, vp, vp, vp, i32, vp, sz, vp]),
    "segmi_seg_metrics": (i32, [vp, i32, vp, i64, i32, vp, vp]),
    "segmi_sgd_chunk_elems": (i32, []),
    "segmi_sgd_step": (i32, [vp, i32, vp, vp, vp, i32, vp]),
    "segmi_sgd_hyper_floats": (i32, []),
    "segmi_sgd_step_dev": (i32, [vp, i32, vp, vp]),
    "segmi_pyramid_pool_workspace": (sz, [i32, i32, i32, i32, i32, vp]),
    "segmi_pyramid_pool_fwd": (i32, [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]),
    "segmi_pyramid_pool_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsegmi.so is not built (%s). Run `python pytorch-segmentation_amd/build.py` "
            "(or __graft_entry__.build()). There is no CPU / eager fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class SegmiError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        raise SegmiError("%s failed: %s (status %d)" % (what or "segmi call", lib.segmi_strerror(rc).decode(), rc))
