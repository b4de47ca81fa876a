"""ctypes loader of libdynamask_hip.so (the C ABI of include/dynamask_hip.h)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DYNAMASK_HIP_LIB') or os.path.join(_HERE, 'libdynamask_hip.so')      # override: kernel experiments
ABI_VERSION = 27
REQUIRED_BUILD_FLAG = '-packed-fp32-ops'        # dynamask_amd/build.py NO_PACKED_FP32; dm_build_info() must carry it

_c_int = ctypes.c_int
_c_float = ctypes.c_float
_vp = ctypes.c_void_p

# name -> argtypes  (every symbol declared in include/dynamask_hip.h)
SIGNATURES = {
    'dm_error_string': ([_c_int], ctypes.c_char_p),
    'dm_abi_version': ([], _c_int),
    'dm_build_info': ([], ctypes.c_char_p),
    'dm_reload_env_knobs': ([], _c_int),
    'dm_roi_align_fwd': ([_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_float, _vp, _vp, _vp], _c_int),
    'dm_roi_align_workspace_bytes': ([_c_int, _c_int], ctypes.c_longlong),
    'dm_roi_align_fwd_ws': ([_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_float, _vp, _vp, _vp, ctypes.c_longlong, _vp], _c_int),
    'dm_roi_align_bwd': ([_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _c_float, _vp], _c_int),
    'dm_conv_packed_cout': ([_c_int], _c_int),
    'dm_conv_packed_floats': ([_c_int, _c_int, _c_int, _vp], ctypes.c_longlong),
    'dm_conv_pack_weight': ([_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp], _c_int),
    'dm_conv_pack_weight_batch': ([_vp, _c_int, _vp], _c_int),
    'dm_conv2d_fwd': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp], _c_int),
    'dm_conv2d_fwd_ws': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, ctypes.c_longlong, _vp], _c_int),
    'dm_conv2d_splitk_floats': ([_c_int, _c_int, _c_int, _c_int, _c_int], ctypes.c_longlong),
    'dm_conv2d_fwd_masked': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_conv1x1_group_fwd': ([_c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _vp, _vp], _c_int),
    'dm_point_sample_fwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_float, _vp, _vp], _c_int),
    'dm_class_logits_fwd': ([_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp], _c_int),
    'dm_class_logits_up2x_fwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp], _c_int),
    'dm_deform_conv_fwd': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_deform_conv_fwd_ws': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_int, _vp, _vp, ctypes.c_longlong, _vp], _c_int),
    'dm_deform_conv_tout_supported': ([_c_int, _c_int, _c_int, _c_int, _c_int, _c_int], _c_int),
    'dm_deform_conv_tout_fwd': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, _c_int, _vp, _c_int, _vp, _vp], _c_int),
    'dm_deform_conv_splitk_floats': ([_c_int, _c_int, _c_int, _c_int, _c_int], ctypes.c_longlong),
    'dm_upsample2x_bilinear_fwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_boundary_merge': ([_vp, _vp, _c_int, _c_int, _vp], _c_int),
    'dm_boundary_merge_chain': ([_vp, _vp, _vp, _vp, _c_int, _c_int, _vp], _c_int),
    'dm_stage_head_fwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_float, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp], _c_int),
    'dm_deconv_pack_weight': ([_vp, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_deconv2x2_fwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_carafe_fwd': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_gumbel_select_fwd': ([_vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp, _vp], _c_int),
    'dm_gumbel_select_bwd': ([_vp, _vp, _c_int, _c_int, _c_float, _vp, _vp], _c_int),
    'dm_class_balance_fwd_bwd': ([_vp, _c_int, _c_int, _vp, _vp, _vp], _c_int),
    'dm_bn_stats': ([_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_float, _vp, _vp, _vp], _c_int),
    'dm_bn_scratch_floats': ([_c_int], ctypes.c_longlong),
    'dm_bn_relu_maxpool_fwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_float, _vp, _vp], _c_int),
    'dm_bn_relu_maxpool_argmax': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_float, _vp, _vp], _c_int),
    'dm_relu_bwd': ([_vp, _vp, ctypes.c_longlong, _vp], _c_int),
    'dm_sigmoid_bwd': ([_vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _vp, ctypes.c_longlong, _c_int, _c_int, _vp, _c_int, _vp], _c_int),
    'dm_channel_sum': ([_vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _vp, _c_int, _vp], _c_int),
    'dm_conv2d_wgrad_scratch_floats': ([], ctypes.c_longlong),
    'dm_conv2d_wgrad_slab': ([_vp, ctypes.c_longlong, _c_int, _vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp, ctypes.c_longlong, _vp], _c_int),
    'dm_conv2d_wgrad': ([_vp, ctypes.c_longlong, _c_int, _vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_upsample2x_bilinear_bwd': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_point_sample_bwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_float, _vp, _vp], _c_int),
    'dm_class_logits_bwd': ([_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_class_logits_bwd_slab': ([_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp], _c_int),
    'dm_class_logits_bwd_scratch_floats': ([_c_int, _c_int], ctypes.c_longlong),
    'dm_deform_im2col': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_deform_col2im_coord': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp], _c_int),
    'dm_deform_coord_grad': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_deform_col2im': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_dcn_weight_permute': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp], _c_int),
    'dm_bn_relu_maxpool_bwd': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_float, _vp, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_rle_scratch_ints': ([_c_int, _c_int, _c_int], ctypes.c_longlong),
    'dm_rle_encode_canvas': ([_vp, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp], _c_int),
    'dm_paste_rle': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp], _c_int),
    'dm_rle_string': ([_vp, _c_int, ctypes.c_longlong, ctypes.c_char_p, ctypes.c_longlong], ctypes.c_longlong),
    'dm_bbox_decode': ([_vp, _c_int, _c_int, _vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _c_float, _c_float, _c_float, _c_float, _c_float, _vp, _vp, _vp], _c_int),
    'dm_nms_mask': ([_vp, _c_int, _c_float, _c_int, _vp, _vp], _c_int),
    'dm_nms_reduce': ([_vp, _c_int, _vp, _c_int], _c_int),
    'dm_fc_scratch_floats': ([_c_int, _c_int, _c_int], ctypes.c_longlong),
    'dm_fc_fwd': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp], _c_int),
    'dm_sgd_momentum_step': ([_vp, _vp, _vp, ctypes.c_longlong, _c_float, _c_float, _c_float, _c_float, _c_int, _vp], _c_int),
    'dm_mask_target_rois': ([_vp, _vp, _c_int, _c_float, _c_float, _vp, _vp], _c_int),
    'dm_polygon_mask_targets': ([_vp, _vp, _vp, _c_int, _vp, _vp, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_threshold_ge': ([_vp, ctypes.c_longlong, _c_float, _vp, _vp], _c_int),
    'dm_paste_masks': ([_vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_float, _c_int, _vp, _vp], _c_int),
    'dm_detail_target': ([_vp, _c_int, _c_int, _c_float, _c_float, _vp, _vp, _vp], _c_int),
    'dm_carafe_bwd_scratch_floats': ([_c_int, _c_int, _c_int, _c_int, _c_int], ctypes.c_longlong),
    'dm_carafe_bwd': ([_vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _vp, _vp, _vp], _c_int),
    'dm_upsample2x_nearest_fwd': ([_vp, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_upsample2x_nearest_bwd': ([_vp, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_pixel_unshuffle2x': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_bbox_overlaps': ([_vp, _c_int, _vp, _c_int, _c_int, _c_float, _vp, _vp], _c_int),
    'dm_max_iou_assign': ([_vp, _c_int, _c_int, _c_float, _c_float, _c_float, _c_float, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_random_sample': ([_vp, _vp, _c_int, _c_int, _vp, _c_int, _vp, _vp, _vp, _c_int, _c_int, _c_int, ctypes.c_double, _vp,
                          _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_ignore_columns': ([_vp, _c_int, _c_int, _vp, _c_int, _c_int, _c_float, _vp], _c_int),
    'dm_bbox_encode': ([_vp, _vp, _c_int, _vp, _vp, _vp, _vp], _c_int),
    'dm_softmax_ce_fwd_bwd': ([_vp, _vp, _vp, _c_int, _c_int, _c_float, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_l1_loss_fwd_bwd': ([_vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_float, _vp, _vp, _vp, _vp], _c_int),
    'dm_sumsq_scratch_floats': ([], ctypes.c_longlong),
    'dm_sumsq': ([_vp, ctypes.c_longlong, _vp, _vp, _vp], _c_int),
    'dm_clip_scale': ([_vp, ctypes.c_longlong, _vp, _c_float, _vp], _c_int),
    'dm_scale': ([_vp, ctypes.c_longlong, _c_float, _vp], _c_int),
    'dm_mask_loss_scratch_floats': ([_c_int], ctypes.c_longlong),
    'dm_mask_loss_fwd_bwd': ([_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _vp, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_mask_loss_stage': ([_vp, _vp, _vp, _vp, _vp, _c_int, _c_int, _c_int, _c_int, _c_float, _vp, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_conv2d_wgrad_fx': ([_vp, ctypes.c_longlong, _c_int, _vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_channel_sum_fx': ([_vp, ctypes.c_longlong, _c_int, _c_int, _c_int, _vp, _vp], _c_int),
    'dm_class_logits_bwd_fx': ([_vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int, _vp, _vp, _vp, _vp, _vp], _c_int),
    'dm_point_sample_bwd_fx': ([_vp, _c_int, _c_int, _c_int, _c_int, _vp, _c_int, _c_int, _c_float, _vp, _vp], _c_int),
    'dm_fx_to_float': ([_vp, ctypes.c_longlong, _vp, _c_int, _c_int, _vp], _c_int),
}

class PackJob(ctypes.Structure):
    """dm_pack_job of include/dynamask_hip.h."""
    _fields_ = [('w', ctypes.c_void_p), ('w_packed', ctypes.c_void_p), ('Cout', _c_int), ('Cin', _c_int), ('ksize', _c_int),
                ('transpose_flip', _c_int), ('num_srcs', _c_int), ('src_channels', _c_int * 4), ('ld', _c_int), ('c0', _c_int)]


_LIB = None
_PROXY = None       # hazard.wrap_lib(_LIB), handed out while the stream-hazard tracker is on (DM_HAZARD, hazard.ENABLED)


class DynaMaskLibraryError(RuntimeError):
    pass


def check_build_info(info):
    """Refuse a library that does not say it was compiled without packed fp32 (include/dynamask_hip.h dm_build_info)."""
    if REQUIRED_BUILD_FLAG not in info and os.environ.get('DM_ALLOW_PACKED_FP32') != '1':
        raise DynaMaskLibraryError(
            f'{LIB_PATH} was not built with {REQUIRED_BUILD_FLAG!r} (dm_build_info: {info!r}): packed fp32 '
            'instructions dropped a product under multi-queue load (profiles/r05_race_hunt.txt).  Rebuild with '
            '`python -m dynamask_amd.build`, or set DM_ALLOW_PACKED_FP32=1 for an A/B measurement')


def lib():
    """Load the library once.  Fails loudly: there is no fallback path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DynaMaskLibraryError(
                f'{LIB_PATH} not found: build it with `python -m dynamask_amd.build` '
                '(or __graft_entry__.build()); dynamask_amd has no CPU/eager fallback')
        # PyTorch-ROCm ships its own libamdhip64; import it FIRST so that the
        # process has exactly one HIP runtime (the library's DT_NEEDED
        # libamdhip64.so.7 then binds to the copy torch already loaded, and the
        # streams / device pointers torch hands us belong to the same runtime).
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (argtypes, restype) in SIGNATURES.items():
            fn = getattr(L, name)       # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = restype
        if L.dm_abi_version() != ABI_VERSION:
            raise DynaMaskLibraryError('libdynamask_hip.so ABI version mismatch: rebuild')
        check_build_info(L.dm_build_info().decode())
        _LIB = L
    from . import hazard
    if hazard.ENABLED[0]:
        global _PROXY
        if _PROXY is None:
            _PROXY = hazard.wrap_lib(_LIB)
        return _PROXY
    return _LIB


def check(rc, what):
    if rc != 0:
        msg = lib().dm_error_string(rc).decode()
        raise RuntimeError(f'{what} failed: {msg} (code {rc})')
