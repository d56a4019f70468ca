"""ctypes binding of libmdt_hip.so (C ABI declared in include/mdt_hip.h).

The library is the product: there is NO fallback.  If it is missing or fails to
load, every op raises ("fail loudly").
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_longlong, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmdt_hip.so")

MDT_OK = 0
MDT_ERR_UNSUPPORTED = -4
NMS_RULE_GT = 0  # GPU rule, IoU >  thresh (nms_kernel.cu:71)
NMS_RULE_GE = 1  # CPU rule, IoU >= thresh (nms.c:64)

_lib = None
# tuning hooks: exported by libmdt_hip_tuning.so only (include/mdt_hip_ab.h), bound by use_tuning_build()
_TUNING_SIGNATURES = {
    "mdt_debug_bwd3": (None, [c_void_p, c_int, c_int]),
    "mdt_debug_fwd_stamps": (None, [c_void_p]),
}

_SIGNATURES = {
    "mdt_version": (c_char_p, []),
    "mdt_error_string": (c_char_p, [c_int]),
    "mdt_crop_and_resize_3d_forward": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_float, c_void_p, c_void_p]),
    "mdt_crop_and_resize_3d_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p]),
    "mdt_crop_and_resize_2d_forward_bf16": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "mdt_crop_and_resize_3d_forward_u8": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p]),
    "mdt_crop_and_resize_2d_forward_u8": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "mdt_crop_and_resize_backward_workspace_bytes": (c_size_t, [c_int] * 9),
    "mdt_crop_and_resize_3d_backward": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_pyramid_roi_align_forward": (c_int, [c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    "mdt_pyramid_roi_align_forward_cl": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_void_p]),
    "mdt_pyramid_roi_align_backward": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdt_pyramid_roi_align_backward_accumulate": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdt_upsample2x_yx_cl_forward": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_longlong, c_void_p]),
    "mdt_upsample2x_yx_cl_backward": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_int, c_longlong, c_void_p]),
    "mdt_conv3x3x3_small_supported": (c_int, [c_int] * 5),
    "mdt_conv3x3x3_small_forward": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mdt_conv3x3x3_small_forward_bias_act": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mdt_conv_stem_wgrad_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mdt_conv_stem_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 11 + [c_void_p, c_size_t, c_void_p]),
    "mdt_s2d221_input": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mdt_s2d221_fold_input_grad": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    "mdt_conv_s221_wgrad_supported": (c_int, [c_int] * 7),
    "mdt_conv_s221_wgrad_workspace_bytes": (c_size_t, [c_int] * 7),
    "mdt_conv_s221_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    "mdt_conv_s221_input_grad_supported": (c_int, [c_int] * 6),
    "mdt_conv_s221_input_grad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "mdt_conv_win_wgrad_supported": (c_int, [c_int] * 7),
    "mdt_conv_win_wgrad_workspace_bytes": (c_size_t, [c_int] * 7),
    "mdt_conv_win_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    "mdt_conv_win_forward_supported": (c_int, [c_int] * 6),
    "mdt_conv_win_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 7 + [c_void_p]),
    "mdt_conv_s221_forward_supported": (c_int, [c_int] * 6),
    "mdt_conv_s221_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 7 + [c_void_p]),
    "mdt_conv1x1_dgrad_add_supported": (c_int, [c_int, c_int]),
    "mdt_conv1x1_dgrad_add": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p]),
    "mdt_adam_flat": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong] + [c_double] * 5 + [c_longlong, c_double, c_void_p]),
    "mdt_adam_flat_segments_workspace_bytes": (c_size_t, [c_int]),
    "mdt_adam_flat_segments": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int]
                               + [c_double] * 6 + [c_void_p, c_size_t, c_void_p]),
    "mdt_conv_stem_forward_supported": (c_int, [c_int] * 7),
    "mdt_conv_stem_forward": (c_int, [c_void_p] * 4 + [c_int] * 12 + [c_void_p]),
    "mdt_conv3x3x3_small_wgrad_workspace_bytes": (c_size_t, [c_int] * 5),
    "mdt_conv3x3x3_small_wgrad": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, c_size_t, c_void_p]),
    "mdt_conv1x1_wgrad_workspace_bytes": (c_size_t, [ctypes.c_longlong, c_int, c_int]),
    "mdt_conv1x1_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mdt_crop_and_resize_3d_backward_ordered": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p]),
    "mdt_crop_and_resize_2d_forward": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_float, c_void_p, c_void_p]),
    "mdt_crop_and_resize_2d_backward": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_crop_and_resize_2d_backward_ordered": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p]),
    "mdt_maxpool3d_k3s221_cl_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_maxpool3d_k3s221_cl_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_filter_flip_transpose": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_filter_flip_transpose_batched": (c_int, [c_void_p, c_int, c_longlong, c_void_p]),
    "mdt_bias_act_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p]),
    "mdt_bias_act_backward_workspace_bytes": (c_size_t, [c_longlong, c_int, c_longlong]),
    "mdt_bias_act_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p, c_size_t, c_void_p]),
    "mdt_conv1x1_forward_supported": (c_int, [c_int, c_int]),
    "mdt_conv1x1_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int, c_void_p]),
    "mdt_conv1x1_backward_supported": (c_int, [c_int, c_int]),
    "mdt_conv1x1_backward_workspace_bytes": (c_size_t, [c_longlong, c_int]),
    "mdt_conv1x1_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mdt_conv_c0_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "mdt_conv_c0_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_conv_c0_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mdt_conv_c0_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mdt_conv_seg_supported": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mdt_conv_seg_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_conv_seg_input_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_conv_seg_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mdt_conv_seg_weight_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "mdt_rpn_heads_forward_supported": (c_int, [c_int, c_int, c_int]),
    "mdt_rpn_heads_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_int, c_longlong, c_longlong, c_void_p]),
    "mdt_bias_act_forward_upsampled_supported": (c_int, [c_int, c_longlong]),
    "mdt_bias_act_forward_upsampled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "mdt_bias_grad_to_channels_last_supported": (c_int, [c_int]),
    "mdt_bias_grad_to_channels_last_workspace_bytes": (c_size_t, [c_int, c_int, c_longlong]),
    "mdt_bias_grad_to_channels_last": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_longlong, c_void_p, c_size_t, c_void_p]),
    "mdt_bias_act_backward_ticket": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_longlong, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "mdt_roi_levels": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mdt_rpn_sample_supported": (c_int, [c_int, c_int, c_int]),
    "mdt_rpn_sample_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mdt_rpn_sample": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 6 + [c_void_p, c_size_t, c_void_p]),
    "mdt_anchor_delta_targets": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p, c_void_p]),
    "mdt_detection_targets_supported": (c_int, [c_int] * 5),
    "mdt_detection_targets": (c_int, [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 8 + [c_int] * 8 + [c_float] * 3 + [c_void_p] * 9),
    "mdt_rpn_patch_gather": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mdt_rpn_patch_move_add": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "mdt_rpn_patch_scatter_add_ordered": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "mdt_rpn_patch_scatter_add": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mdt_refine_detections_supported": (c_int, [c_int, c_int, c_int]),
    "mdt_refine_detections_pre": (c_int, [c_void_p] * 6 + [c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "mdt_refine_detections_post": (c_int, [c_void_p] * 6 + [c_float, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 6),
    "mdt_nms_mask_3d": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdt_nms_mask_2d": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdt_nms_mask_full_3d": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdt_nms_mask_full_2d": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p]),
    "mdt_nms_workspace_bytes": (c_size_t, [c_int]),
    "mdt_nms_3d": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_nms_2d": (c_int, [c_void_p, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_nms_3d_batched": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_nms_2d_batched": (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_decode_clip_boxes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "mdt_generate_anchors": (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "mdt_anchor_match_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mdt_anchor_match": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_anchor_match_batched_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mdt_anchor_match_batched": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_size_t, c_void_p]),
    "mdt_nms_2to3d_workspace_bytes": (c_size_t, [c_int]),
    "mdt_nms_2to3d": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_wbc_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mdt_weighted_box_clustering": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_double, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

# libmdt_hip_ab.so (include/mdt_hip_ab.h): superseded kernel generations, A/B baselines and test subjects ONLY -- nothing under
# medicaldetectiontoolkit_amd/ calls ab_lib() on its own; tests and tools ask for these modes explicitly
AB_LIB_PATH = os.path.join(_HERE, "libmdt_hip_ab.so")
_AB_SIGNATURES = {
    "mdt_crop_and_resize_backward_twophase_workspace_bytes": (c_size_t, [c_int] * 9),
    "mdt_crop_and_resize_3d_backward_twophase": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_crop_and_resize_2d_backward_twophase": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, c_size_t, c_void_p]),
    "mdt_debug_bwd_timestamps": (None, [c_void_p]),
    "mdt_crop_and_resize_3d_backward_atomic": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p]),
    "mdt_ab_crop_and_resize_backward_territory": (c_int, [c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 9 + [c_void_p, c_void_p]),
}
_ab_lib = None


def ab_lib():
    """the A/B library (tests / tools); raises if it was not built"""
    global _ab_lib
    if _ab_lib is None:
        if not os.path.exists(AB_LIB_PATH):
            raise RuntimeError("libmdt_hip_ab.so not found at %s -- make -C medicaldetectiontoolkit_amd/csrc" % AB_LIB_PATH)
        lib()           # (torch's HIP runtime first, see lib())
        handle = ctypes.CDLL(AB_LIB_PATH)
        for name, (res, args) in _AB_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _ab_lib = handle
    return _ab_lib


def use_tuning_build():
    """TOOLS ONLY (tools/bwd3_probe.py, tools/fwd_stamp_probe.py, tools/profile_case.py): make lib() load libmdt_hip_tuning.so -- the product
    sources compiled with -DMDT_TUNING_HOOKS -- instead of libmdt_hip.so, and bind mdt_debug_bwd3 / mdt_debug_fwd_stamps.  Must be called
    before the first lib(); the package never calls it."""
    global LIB_PATH
    if _lib is not None:
        raise RuntimeError("use_tuning_build() must precede the first _lib.lib() call")
    path = os.path.join(_HERE, "libmdt_hip_tuning.so")
    if not os.path.exists(path):
        raise RuntimeError("libmdt_hip_tuning.so not found at %s -- make -C medicaldetectiontoolkit_amd/csrc" % path)
    LIB_PATH = path
    _SIGNATURES.update(_TUNING_SIGNATURES)
    return lib()


def lib():
    """Load (once) and return the ctypes handle; raises RuntimeError if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmdt_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C medicaldetectiontoolkit_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # torch bundles its own libamdhip64; import it FIRST so this library binds to the same HIP runtime
        # (loading ours first pulls in /opt/rocm's copy and kernels then fail with "no ROCm-capable device")
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


# > 0 while training.GraphedTrainStep captures a hipGraph: helpers that keep grown-on-demand device workspaces in module-level caches
# must then hand out a fresh tensor from the graph's private pool instead (a cached workspace that is re-allocated LATER -- a bigger
# shape in an eager call -- would leave the captured kernels pointing at freed memory)
CAPTURING = 0

_COUNTED = None     # {symbol: original ctypes function} while count_calls() is on
CALLS = {}          # symbol -> number of calls since count_calls(True)


def count_calls(enable=True):
    """Test / diagnosis hook: count the calls of every C-ABI entry point by name (`CALLS`), so a test can assert that a use-rule really
    dispatched the kernel it claims (e.g. mdt_conv3x3x3_small_forward inside the assembled training step) instead of silently
    routing back to MIOpen.  Off by default: the ctypes functions are called directly, without a Python wrapper in between."""
    global _COUNTED
    handle = lib()
    if enable and _COUNTED is None:
        CALLS.clear()
        _COUNTED = {}
        for name in _SIGNATURES:
            fn = getattr(handle, name)
            _COUNTED[name] = fn

            def counted(*a, _fn=fn, _name=name):
                CALLS[_name] = CALLS.get(_name, 0) + 1
                return _fn(*a)
            setattr(handle, name, counted)
    elif not enable and _COUNTED is not None:
        for name, fn in _COUNTED.items():
            setattr(handle, name, fn)
        _COUNTED = None


def check(code, what):
    if code != MDT_OK:
        raise RuntimeError("%s failed: %s (code %d)" % (what, lib().mdt_error_string(code).decode(), code))


def raw_stream(t=None):
    """hipStream_t (as an int) of the calling thread's current stream on the device of tensor `t` (default: the current device).
    One C call: `torch.cuda.current_stream().cuda_stream` builds a Stream object and costs ~10 us of host time, and the epilogue /
    convolution helpers ask ~200 times per training step."""
    import torch
    idx = t.device.index if t is not None else None
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)


def current_stream_ptr():
    return c_void_p(raw_stream())


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU: the MI355X HIP path is the only implementation "
                           "(no CPU fallback in the product path)" % name)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
