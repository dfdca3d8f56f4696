"""ctypes binding of libdtt_hip.so (the C ABI declared in include/dtt_hip.h).

There is no fallback: if the shared library is missing or a call fails, the op raises.  PyTorch is
used only as plumbing here (device memory, current stream).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("DTT_HIP_LIBRARY") or os.path.join(PKG_ROOT, "lib", "libdtt_hip.so")   # env override: developer A/B builds
CSRC = os.path.join(PKG_ROOT, "csrc")

_P = ctypes.c_void_p
_I = ctypes.c_int
_F = ctypes.c_float
_L = ctypes.c_long
_Z = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/dtt_hip.h one to one
SIGNATURES = {
    "dtt_abi_version": (_I, []),
    "dtt_last_error": (ctypes.c_char_p, []),
    "dtt_profile_attach": (_I, [ctypes.c_char_p, _P, _P, _I]),
    "dtt_profile_count": (_I, []),
    "dtt_correlation_output_shape": (_I, [_I] * 8 + [ctypes.POINTER(_I)] * 3),
    "dtt_correlation_forward_workspace_bytes": (_Z, [_I] * 9),
    "dtt_correlation_forward": (_I, [_P, _I, _I, _I, _I, _L, _P, _I, _I, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    "dtt_correlation_forward_strided": (_I, [_P, _I, _I, _I, _I, _L, _L, _L, _P, _I, _I, _I, _P, _P, _Z, _I, _I, _I, _I, _I, _I, _P]),
    "dtt_correlation_forward_nhwc": (_I, [_P, _I, _I, _I, _I, _L, _L, _L, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dtt_correlation_nhwc_plan_check": (_I, [_I, _I, _I, _I, _I]),
    "dtt_correlation_nhwc_plan": (_I, [_I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dtt_correlation_backward": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dtt_correlation_backward_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "dtt_correlation_backward_nhwc_strided": (_I, [_P, _L, _L, _L, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I,
                                                   _P, _Z, _P]),
    "dtt_correlation_backward_nhwc_phase": (_I, [_P, _L, _L, _L, _I, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I,
                                                 _P, _Z, _P]),
    "dtt_correlation_backward_workspace_bytes": (_Z, [_I] * 9),
    "dtt_correlation_backward_stream_supported": (_I, [_I] * 5),
    "dtt_correlation_backward_plan_check": (_I, [_I] * 6),
    "dtt_correlation_backward_plan": (_I, [_I] * 6 + [_P] * 4),
    "dtt_psroi_pool_forward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "dtt_psroi_pool_backward": (_I, [_P, _P, _I, _I, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dtt_psroi_pool_vote_forward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "dtt_psroi_vote_forward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _P, _P, _P]),
    "dtt_nms_workspace_bytes": (_Z, [_I]),
    "dtt_nms": (_I, [_P, _P, _P, _I, _I, _F, _I, _P, _Z, _P]),
    "dtt_roi_align_forward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "dtt_roi_align_forward_planes": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P]),
    "dtt_roi_align_backward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "dtt_roi_pool_forward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dtt_roi_pool_backward": (_I, [_P, _F, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "dtt_roi_crop_forward": (_I, [_I] * 8 + [_P, _P, _P, _P]),
    "dtt_roi_crop_backward": (_I, [_I] * 8 + [_P, _P, _P, _P, _P]),
    "dtt_proposal_workspace_bytes": (_Z, [_I] * 5),
    "dtt_proposal_forward": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "dtt_proposal_select_sort": (_I, [_P, _I, _I, _I, _I, _I, _P, _Z, _P]),
    "dtt_proposal_decode_nms": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "dtt_anchor_target_assign": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P, _P, _P]),
    "dtt_anchor_target_disable": (_I, [_P, _P, _P, _I, _I, _P]),
    "dtt_proposal_target_assign": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
    "dtt_proposal_target_sample": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "dtt_tracking_target": (_I, [_P, _P, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "dtt_class_nms": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _I, _P, _P, _P]),
    "dtt_bias_act_inplace": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dtt_bias_act_nhwc_inplace": (_I, [_P, _P, _P, _L, _I, _I, _P]),
    "dtt_transpose_batched": (_I, [_P, _P, _I, _I, _I, _P]),
    "dtt_scale_rows_batch": (_I, [_I, _P, _P, _P, _P, _P, _P]),
    "dtt_gather_column_blocks": (_I, [_P, _L, _P, _L, _L, _I, _L, _I, _P]),
    "dtt_maxpool3s2_bias_relu_nhwc": (_I, [_P, _P, _P, _I, _I, _I, _I, _P]),
    "dtt_gemm_batched": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _Z, _P]),
    "dtt_winograd_tiles": (_L, [_I, _I, _I, _I, _I]),
    "dtt_winograd_input_transform": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "dtt_winograd_output_transform": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "dtt_gemm_bias_act": (_I, [_P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _Z, _P]),
    "dtt_gemm_tune": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _Z, _P]),
    "dtt_gemm_batched_tune": (_I, [_P, _P, _P, _I, _L, _I, _I, _P, _Z, _P]),
    "dtt_head_gemm": (_I, [_P, _L, _I, _I, _P, _P, _I, _P, _L, _I, _I, _P]),
    "dtt_psroi_pm_backward": (_I, [_P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _L, _P, _P, _P]),
    "dtt_psroi_pm_backward_heads": (_I, [_P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _L, _I, _P, _I, _I, _P, _P]),
    "dtt_rpn_head_gemm": (_I, [_P, _L, _I, _I, _I, _P, _P, _I, _I, _P, _P, _P]),
    "dtt_rpn_head_grad_rows": (_I, [_P, _P, _P, _I, _I, _I, _P, _L, _I, _P]),
    "dtt_rpn_loss_workspace_bytes": (_Z, [_I, _I]),
    "dtt_rpn_loss_forward": (_I, [_P] * 6 + [_I, _I, _I, _I, _F, _P, _P, _P, _Z, _P]),
    "dtt_rpn_loss_backward": (_I, [_P] * 8 + [_I, _I, _I, _I, _F, _P, _P, _P]),
    "dtt_head_gemm_dw_workspace_bytes": (_Z, [_I, _I, _I]),
    "dtt_head_gemm_dw": (_I, [_P, _L, _I, _P, _L, _I, _I, _I, _P, _P, _Z, _P]),
    "dtt_psroi_pm_forward": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _P, _F, _I, _P, _P, _P]),
    "dtt_psroi_pm_det_forward": (_I, [_P, _L, _I, _I, _I, _I, _I, _I, _P, _F, _I, _I, _P, _P, _P, _P]),
    "dtt_tube_link_workspace_bytes": (_Z, [_I, _I]),
    "dtt_tube_link": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "dtt_anchor_target_finish": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _P, _P, _P, _P, _P]),
    "dtt_anchor_target_device": (_I, [_P, _P, _P, _P] + [_I] * 8 + [_F, _F, _I, _F, _F] + [_P] * 9 + [_P]),
}

_lib = None


class DttLibraryError(RuntimeError):
    pass


def lib():
    """Load libdtt_hip.so (once).  Raises DttLibraryError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DttLibraryError(
                "libdtt_hip.so not found at %s -- build it with `make -C %s` (or __graft_entry__.build()); "
                "there is no CPU / PyTorch fallback for the D&T hot-path ops" % (LIB_PATH, CSRC))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header / library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error():
    msg = lib().dtt_last_error()
    return msg.decode() if msg else ""


def check(status, what):
    """Reference launchers return 1 on success / 0 on failure -> THError('aborting') (correlation_cuda.c:87-89)."""
    if status != 1:
        raise RuntimeError("%s failed: %s" % (what, last_error()))


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def stream_ptr(device=None):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("dtt ops run on the GPU only (got a %s tensor); there is no CPU fallback" % t.device)


def require_f32_contig(name, t):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        # the reference asserts contiguity too (correlation/functions/correlation.py:21-22)
        raise ValueError("%s must be contiguous" % name)
