"""ctypes binding of libsipmask_hip.so (the C ABI declared in include/sipmask_hip.h).

The product path has NO fallback: if the shared library is missing or a call
returns a non-zero status a RuntimeError is raised.  Build with
``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C sipmask_amd/csrc``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsipmask_hip.so")
SM_MAX_LEVELS = 5

SM_CONV_RELU = 1
SM_CONV_OUT_F32 = 2
SM_CONV_RES_ADD = 4
SM_CONV_RES_NEAREST = 8
SM_CONV_IN_RELU = 16
SM_CONV_RELU_NCH = 32
SM_CONV_DBG_DEFORM_GATHER = 0x80000000   # the global-gather deformable loader instead of deform_patch.hip
SM_CONV_DBG_BIG_TILES = 0x04000000
SM_CONV_DBG_TILE256 = 0x00400000
SM_CONV_DBG_HAND_PLACED = 0x00040000
SM_CONV_DBG_PATCH_UNIFORM = 0x00004000
SM_CONV_DBG_LDS_EPILOGUE = 0x01000000
SM_CONV_F16 = 0x00020000                 # IEEE binary16 operands (the x3 head plan), f32 output
SM_CONV_OUT_X3 = 64                      # ... or the next layer's split operand [hi | lo | hi] (forward descriptors)
# backward descriptors only (sm_conv2d_bwd / sm_deform_conv2d_bwd read these bits; the forward entry points never do)
SM_CONV_BWD_GX_BF16 = 64
SM_CONV_BWD_WGRAD_GEMM = 128
SM_CONV_BWD_WGRAD_DIRECT = 256
SM_CONV_BWD_WGRAD_TILE128 = 512
SM_CONV_BWD_DX_SCATTER = 1024            # A/B: d(x) of FeatureAlign's shape by the atomic scatter alone

_i32x5 = C.c_int32 * SM_MAX_LEVELS
_i64x5 = C.c_int64 * SM_MAX_LEVELS
_f32x5 = C.c_float * SM_MAX_LEVELS


class ConvDesc(C.Structure):
    """sm_conv_desc"""
    _fields_ = [
        ("nlev", C.c_int32), ("batch", C.c_int32),
        ("in_h", _i32x5), ("in_w", _i32x5), ("out_h", _i32x5), ("out_w", _i32x5),
        ("in_row0", _i64x5), ("out_row0", _i64x5),
        ("cin", C.c_int32), ("cout", C.c_int32), ("cout_pad", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dil", C.c_int32),
        ("in_cstride", C.c_int32), ("out_cstride", C.c_int32), ("out_coff", C.c_int32),
        ("res_cstride", C.c_int32),
        ("res_h", _i32x5), ("res_w", _i32x5), ("res_row0", _i64x5),
        ("flags", C.c_uint32), ("scale_nch", C.c_int32), ("level_scale", _f32x5),
        ("deform_groups", C.c_int32), ("w_batch_stride", C.c_int64),
        ("ngroups", C.c_int32), ("x_group_rows", C.c_int64), ("y_group_rows", C.c_int64), ("w_group_stride", C.c_int64),
        ("bias_group_stride", C.c_int64), ("gn_group_stride", C.c_int64),
        ("acc_scale", C.c_float),
        ("w_level_stride", C.c_int64), ("bias_level_stride", C.c_int64),
        ("patch_cout_tile", C.c_int32), ("x3_pairs", C.c_int32),
    ]


class ConvPlan(C.Structure):
    """sm_conv_plan"""
    _fields_ = [
        ("lds_dma", C.c_int32), ("k_step", C.c_int32), ("k_padded", C.c_int32), ("tile_cout", C.c_int32),
        ("tile_pos", C.c_int32), ("threads", C.c_int32), ("k_loop", C.c_int32), ("warp_spec", C.c_int32),
        ("blocks", C.c_int64), ("split_k", C.c_int32), ("ring_stages", C.c_int32), ("workspace_bytes", C.c_int64),
    ]


class DetDesc(C.Structure):
    """sm_det_desc"""
    _fields_ = [
        ("batch", C.c_int32), ("nlev", C.c_int32), ("num_classes", C.c_int32),
        ("h", _i32x5), ("w", _i32x5), ("stride", _i32x5), ("row0", _i64x5),
        ("cls_cstride", C.c_int32), ("cls_coff", C.c_int32),
        ("cof_cstride", C.c_int32), ("cof_coff", C.c_int32),
        ("reg_cstride", C.c_int32), ("nms_pre", C.c_int32),
        ("img_h", C.c_int32), ("img_w", C.c_int32), ("kmax", C.c_int32),
        ("scale_factor", C.c_float * 4), ("rescale", C.c_int32), ("reg_prescaled", C.c_int32),
        ("per_image", C.c_void_p),
    ]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float

# name -> (restype, argtypes); every symbol of include/sipmask_hip.h
PROTOTYPES = {
    "sm_version": (_I, []),
    "sm_strerror": (C.c_char_p, [_I]),
    "sm_conv_cout_tile": (_I, [_I]),
    "sm_conv_plan_query": (_I, [C.POINTER(ConvDesc), _I, _I, C.POINTER(ConvPlan)]),
    "sm_conv2d": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "sm_conv2d_ws": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "sm_conv3x3_patch_supported": (_I, [C.POINTER(ConvDesc)]),
    "sm_conv3x3_patch_tiles": (C.c_int64, [C.POINTER(ConvDesc)]),
    "sm_conv3x3_patch_plan": (_I, [C.POINTER(ConvDesc), _P]),
    "sm_conv3x3_patch": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "sm_deform_conv2d": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "sm_deform_conv_window_plan": (_I, [C.POINTER(ConvDesc), _P]),
    "sm_conv2d_gn_stats": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_conv2d_f32": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "sm_deform_conv2d_x3_supported": (_I, [C.POINTER(ConvDesc)]),
    "sm_deform_conv2d_x3_plan": (_I, [C.POINTER(ConvDesc), _P]),
    "sm_deform_conv2d_x3": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "sm_nchw_f32_to_nhwc_f32": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sm_maxpool3x3s2_f32": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sm_groupnorm_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _P]),
    "sm_groupnorm_apply": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _P]),
    "sm_offset_linear": (_I, [_P, _I, _P, _I, _P, _P, _P, _I, _P, _P]),
    "sm_offset_linear_bwd_workspace": (C.c_int64, [C.c_int64, _I]),
    "sm_offset_linear_bwd": (_I, [_P, _I, _P, _I, C.c_int64, _P, _P, _P]),
    "sm_relu_bf16": (_I, [_P, _P, C.c_int64, _P]),
    "sm_copy_segments": (_I, [_I, _P, _P, _P, _P]),
    "sm_bottleneck_tail_supported": (_I, [_I]),
    "sm_bottleneck_tail": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_conv1x1_pair": (_I, [C.c_int64, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_bottleneck_tail_ds": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "sm_split3_f16": (_I, [_P, _I, C.c_int64, _I, _I, _P, _I, _I, _P]),
    "sm_split2_f16": (_I, [_P, C.c_int64, _I, _I, _P, _I, _I, _P]),
    "sm_upsample_bilinear_x3": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sm_upsample_sum2": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sm_gn_stats_f32_fix": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _P]),
    "sm_groupnorm_apply_x3": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _P, _P, _P]),
    "sm_groupnorm_apply_x3p": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _P, _P, _P]),
    "sm_split_pairs_f16": (_I, [_P, _I, C.c_int64, _I, _I, _P, _I, _I, _P]),
    "sm_det_boxes_override": (_I, [_P, _I, _P, _I, _I, _P, _P, _P]),
    "sm_groupnorm": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _P]),
    "sm_maxpool3x3s2": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sm_stem_fused": (_I, [_P, _P, _P, _P, _I, _I, _I, _P]),
    "sm_conv3x3_smallco_supported": (_I, [_P]),
    "sm_conv3x3_smallco": (_I, [_P, _P, _P, _P, _P, _P]),
    "sm_nchw_f32_to_nhwc_bf16": (_I, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "sm_upsample_bilinear": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sm_det_select_workspace": (C.c_int64, [C.POINTER(DetDesc)]),
    "sm_det_select": (_I, [C.POINTER(DetDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_multiclass_nms_workspace": (C.c_int64, [_I, _I, _I]),
    "sm_multiclass_nms": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P, _P]),
    "sm_fast_nms": (_I, [_P, _P, _P, _P, _I, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _P]),
    "sm_deform_conv2d_bwd_workspace": (C.c_int64, [_P]),
    "sm_deform_conv2d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_mask_loss_fwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "sm_mask_loss_bwd": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "sm_track_gather": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    "sm_track_clip": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _I, _P, _P, _P]),
    "sm_track_match": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P]),
    "sm_mask_rescore": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "sm_conv2d_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_wgrad_direct_supported": (_I, [C.POINTER(ConvDesc)]),
    "sm_wgrad_direct_preferred": (_I, [C.POINTER(ConvDesc)]),
    "sm_wgrad_direct": (_I, [C.POINTER(ConvDesc), _P, _P, _P, _P]),
    "sm_weight_prep": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P]),
    "sm_weight_prep_multi": (_I, [_P, _P, _I, _I, _P]),
    "sm_wgrad_finish": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "sm_relu_bwd_bf16": (_I, [_P, _P, _P, C.c_int64, _P]),
    "sm_bias_grad_rows": (_I, [_P, C.c_int64, _I, _I, _P, _P]),
    "sm_gn_bwd_rows": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P, _P]),
    "sm_upsample_bilinear_bwd_rows": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sm_nearest_bwd_rows": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sm_scatter_stride_rows": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "sm_groupnorm_nchw_fwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "sm_groupnorm_nchw_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sm_upsample_bilinear_nchw_fwd": (_I, [_P, _P, C.c_int64, _I, _I, _I, _P]),
    "sm_upsample_bilinear_nchw_bwd": (_I, [_P, _P, C.c_int64, _I, _I, _I, _P]),
    "sm_sgd_step": (_I, [_P, _P, _P, C.c_int64, _F, _F, _F, _I, _P]),
    "sm_sgd_multi": (_I, [_P, _P, _I, _F, _I, _P]),
    "sm_pairs_select_workspace": (C.c_int64, [_P]),
    "sm_pairs_select": (_I, [_P, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "sm_preprocess_u8": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P]),
    "sm_rle_workspace": (C.c_int64, [_I, _I, _I, _I]),
    "sm_mask_rects": (_I, [_P, _I, _I, _F, _F, _F, C.c_double, C.c_double, _P, _P, _P]),
    "sm_rle_encode": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "sm_rle_encode_images": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "sm_nms_workspace": (C.c_int64, [_I]),
    "sm_nms": (_I, [_P, _I, _F, _P, _P, _P, _P]),
    "sm_mask_assemble": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _F, C.c_double, C.c_double, _F,
                              _P, _P, _P, _P]),
    "sm_mask_assemble_lo_supported": (_I, [_I, _I, _I, C.c_double, C.c_double]),
    "sm_mask_assemble_lo_workspace": (C.c_int64, [_I, _I]),
    "sm_mask_assemble_lo": (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, C.c_double, C.c_double,
                                 _F, _P, _P, _P, _P, _P]),
    "sm_fcos_target": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "sm_crop_split_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sm_crop_split_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "sm_crop_split_gt_fwd": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sm_sigmoid_focal_loss_fwd": (_I, [_P, _P, _P, _I, _I, _F, _F, _P]),
    "sm_sigmoid_focal_loss_bwd": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _P]),
}

_lib = None


def load():
    """Load the shared library (once) and bind every prototype.  Raises RuntimeError when the
    library has not been built -- there is no CPU / eager fallback in the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "sipmask_amd: %s is missing; build it with `make -C sipmask_amd/csrc` "
            "(the HIP extension is mandatory, there is no fallback path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().sm_strerror(status).decode()
        raise RuntimeError("sipmask_hip %s failed: %s (%d)" % (what, msg, status))


def ptr(t):
    """device pointer of a tensor (None -> NULL)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            # mirrors M/mmdet/ops/dcn/deform_conv.py:46-47 (NotImplementedError on CPU tensors)
            raise NotImplementedError("sipmask_amd ops are HIP-only: got a %s tensor" % t.device)
