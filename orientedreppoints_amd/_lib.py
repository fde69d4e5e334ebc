"""ctypes binding of liborp_hip.so (the C ABI declared in include/orp_hip.h).

The product path has NO CPU fallback: if the HIP library is missing the import of any operator fails loudly.
PyTorch is used for device memory, the current stream and (elsewhere) torch.distributed only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORP_HIP_LIB") or os.path.join(_HERE, "csrc", "liborp_hip.so")   # ORP_HIP_LIB: dev aid (instrumented builds)

ORP_OK, ORP_EINVAL, ORP_EWORKSPACE, ORP_ETOOBIG = 0, -1, -2, -3
ORP_NMS_MAX_BOXES = 131072

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/orp_hip.h one to one
_SIGNATURES = {
    "orp_version": (ctypes.c_char_p, []),
    "orp_rnms_workspace_bytes": (_sz, [_i]),
    "orp_rnms": (_i, [_vp, _i, _f, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "orp_rnms_batched_workspace_bytes": (_sz, [_i, _i, _i]),
    "orp_rnms_batched": (_i, [_vp, _i, _vp, _i, _i, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "_poly_nms": (None, [_vp, _vp, _vp, _i, _i, _f, _i]),
    "_overlaps": (None, [_vp, _vp, _vp, _i, _i, _i]),
    "orp_quad_iou_matrix": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp]),
    "orp_poly_overlaps": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "orp_box_iou_rotated": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "orp_box_iou_rotated_host": (_i, [_vp, _i, _vp, _i, _vp]),
    "orp_minarearect": (_i, [_vp, _i, _vp, _vp]),
    "orp_minarearect_decode": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
    "orp_libm_eval": (_i, [_vp, _vp, ctypes.c_long, _i, _vp, _vp]),
    "orp_convex_iou": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "orp_convex_giou": (_i, [_vp, _vp, _i, _vp, _vp]),
    "orp_points_justify": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "orp_points_in_quad_aligned": (_i, [_vp, _vp, _i, _vp, _vp]),
    "orp_chamfer2d_forward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "orp_chamfer2d_backward": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "orp_sigmoid_focal_loss_forward": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "orp_sigmoid_focal_loss_backward": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "orp_sigmoid_focal_loss_forward_f64": (_i, [_vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "orp_sigmoid_focal_loss_backward_f64": (_i, [_vp, _vp, _vp, _i, _i, _f, _f, _vp, _vp]),
    "orp_point_assign_workspace_bytes": (_sz, [_i]),
    "orp_point_assign": (_i, [_vp, _i, _vp, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "orp_max_iou_assign_workspace_bytes": (_sz, [_i]),
    "orp_max_iou_assign": (_i, [_vp, _i, _i, _f, _f, _f, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "orp_apaa_feature_dissimilarity": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp]),
    "orp_apaa_select": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_double, _vp, _vp]),
    "orp_pointset_target": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "orp_points_from_offsets": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "orp_gather_levels": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "orp_gather_levels_backward": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp]),
    "orp_outline_samples": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "orp_profile_enable": (_i, [_i]),
    "orp_profile_read": (_i, [_i, _vp, _vp, _i]),
    "orp_dcn_fast_path_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "orp_dcn_pack_weight": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "orp_dcn_packed_weight_floats": (_sz, [_i, _i, _i, _i]),
    "orp_dcn_set_split_mode": (_i, [_i]),
    "orp_dcn_get_split_mode": (_i, []),
    "orp_dcn_forward_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_dcn_forward_multi": (_i, [_vp, _i, _i, _i, _i, _vp] + [_i] * 10 + [_vp, _sz, _vp]),
    "orp_dcn_forward_multi_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp] + [_i] * 11 + [_vp, _sz, _vp]),
    "orp_dcn_forward_pair": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _sz, _vp]),
    "orp_dcn_head_packed_floats": (_sz, []),
    "orp_dcn_pack_head_weight": (_i, [_vp, _i, _vp, _vp]),
    "orp_dcn_forward_pair_heads": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp] + [_i] * 9 + [_vp, _sz, _vp]),
    "orp_dcn_half_path_ok": (_i, [_i, _i, _i, _i, _i, _i]),
    "orp_dcn_pack_weight_h": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _vp]),
    "orp_dcn_forward_h_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_dcn_forward_multi_h": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp] + [_i] * 12 + [_vp, _sz, _vp]),
    "orp_dcn_im2col": (_i, [_vp, _vp, _vp] + [_i] * 13 + [_vp, _vp]),
    "orp_dcn_col2im": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp, _vp, _vp, _vp]),
    "orp_dcn_im2col_f64": (_i, [_vp, _vp, _vp] + [_i] * 13 + [_vp, _vp]),
    "orp_dcn_col2im_f64": (_i, [_vp, _vp, _vp, _vp] + [_i] * 13 + [_vp, _vp, _vp, _vp]),
    "orp_dcn_col2im_nhwc": (_i, [_vp, _vp, _vp] + [_i] * 12 + [_vp, _vp, _vp]),
    "orp_dcn_forward_direct": (_i, [_vp] * 6 + [_i] * 15 + [_vp]),
    "orp_dcn_backward_mfma_ok": (_i, [_i] * 6),
    "orp_dcn_backward_workspace_bytes": (_sz, [_vp] + [_i] * 10),
    "orp_dcn_backward_multi": (_i, [_vp, _i, _i, _i, _i, _vp, _vp] + [_i] * 9 + [_vp, _sz, _vp]),
    "orp_dcn_backward_multi_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp] + [_i] * 9 + [_vp, _sz, _vp]),
    "orp_poly_nms_f64_workspace_bytes": (_sz, [_i]),
    "orp_poly_nms_f64": (_i, [_vp, _i, ctypes.c_double, _vp, _vp, _vp, _sz, _vp]),
    "orp_soft_rnms_host": (_i, [_vp, _i, _f, _i, _f, _f, _vp, _vp]),
    "orp_conv1x1_packed_floats": (_sz, [_i]),
    "orp_conv1x1_ok": (_i, [_i, _i]),
    "orp_conv1x1_pack_weight": (_i, [_vp, _i, _i, _vp, _vp]),
    "orp_conv1x1_multi": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "orp_pp_select_scratch_bytes": (_sz, [_i, _i, _i]),
    "orp_pp_select": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "orp_pp_gather": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "orp_pp_compact_scratch_bytes": (_sz, [_i]),
    "orp_pp_compact": (_i, [_vp, _vp, _i, _i, _i, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "orp_pp_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "orp_groupnorm_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_groupnorm_act_multi": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _i, _vp, _sz, _vp]),
    "orp_groupnorm_act_multi_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "orp_groupnorm_act_multi_nhwc": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "orp_groupnorm_act_multi_train": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _vp]),
    "orp_groupnorm_backward_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_groupnorm_act_multi_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "orp_voc_best_match_f64": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp]),
    "orp_affine_act": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "orp_bias_act_multi": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _vp]),
    "orp_conv3x3_small_ok": (_i, [_i, _i]),
    "orp_conv3x3_small_multi": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "orp_conv3x3_small_multi_ex": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "orp_border_rows": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "orp_giou_rows": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    "orp_segment_finish": (_i, [_vp, _vp, _vp, _i, _i, _vp, _f, _i, _vp, _vp, _vp]),
    "orp_conv3x3_small_workspace_bytes": (ctypes.c_size_t, [_vp, _vp, _i, _i, _i]),
    "orp_conv3x3_small_multi_strided": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, ctypes.c_size_t, _vp]),
    "orp_conv_split_ok": (_i, [_i, _i, _i, _i]),
    "orp_conv_split_multi": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _sz, _vp, _i, _vp]),
    "orp_conv_split_multi_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i] + [_i] * 11 + [_vp, _sz, _vp, _vp]),
    "orp_conv_wgrad_split_ok": (_i, [_i, _i, _i, _i]),
    "orp_conv_wgrad_split_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_conv_wgrad_split": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "orp_nchw_to_nhwc_multi_amax": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp]),
    "orp_groupnorm_act_multi_cl_amax": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "orp_dcn_forward_pair_amax": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp] + [_i] * 11 + [_vp, _sz, _vp, _i, _vp]),
    "orp_nchw_to_nhwc_multi": (_i, [_vp, _i, _i, _i, _vp]),
    "orp_groupnorm_cl_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_groupnorm_act_multi_cl": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp]),
    "orp_debug_amax_log": (_i, [_vp, _i]),
    "orp_conv_split_gn_partial_floats": (_sz, [_vp, _i, _i, _i, _i]),
    "orp_conv_split_multi_gn": (_i, [_vp, _i, _i, _i, _i, _vp, _vp] + [_i] * 7 + [_vp, _i, _vp, _sz, _i, _vp, _sz, _vp, _i, _i, _vp]),
    "orp_conv_split_gn_finish": (_i, [_vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "orp_affine_act_multi_cl": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _i, _i, _vp, _vp]),
}

_lib = None


class OrpHipError(RuntimeError):
    pass


def lib():
    """Load liborp_hip.so (once).  Raises if it has not been built: there is deliberately no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrpHipError(
                "orientedreppoints_amd: HIP library %s is missing. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the hot path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)          # AttributeError = the .so is stale vs the header: also loud
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what):
    if rc != ORP_OK:
        names = {ORP_EINVAL: "ORP_EINVAL", ORP_EWORKSPACE: "ORP_EWORKSPACE", ORP_ETOOBIG: "ORP_ETOOBIG"}
        raise OrpHipError("%s failed: %s" % (what, names.get(rc, "hipError %d" % rc)))


def ptr(t):
    """Device (or host) address of a contiguous tensor, or NULL."""
    if t is None:
        return None
    assert t.is_contiguous(), "orientedreppoints_amd: tensor must be contiguous"
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    """hipStream_t of PyTorch's current stream on the tensor's device (ops are stream-ordered, never blocking)."""
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def require_cuda(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError("%s must be a CUDA tensor" % name)


_workspaces = {}
_keepalive = None          # list while a GraphedInference capture collects the cached tensors its graph reads


def workspace(device, nbytes):
    """Scratch bytes for ONE op call on PyTorch's current stream of `device`.

    * eager: a cached, growing buffer per (device, stream).  A buffer is only ever used by work queued on the stream it
      is keyed by, so replacing it on growth hands the old block back to the caching allocator in stream order (no
      cross-stream reuse race, no record_stream needed).
    * during a stream capture: a fresh allocation from the capturing graph's private memory pool -- the graph owns its
      scratch for as long as it lives, and nothing an eager call does later (a 20 k-box merge NMS growing the cached
      buffer, a training step, the capacity-overflow fallback) can free or move memory a replay writes to.
    """
    nbytes = max(int(nbytes), 1)
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(device).cuda_stream)
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        if key not in _workspaces and len(_workspaces) >= 32:   # short-lived side streams: drop the oldest buffer only
            _workspaces.pop(next(iter(_workspaces)))           # (its block goes back to the allocator in stream order)
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = w
    return w


def keep_for_graph(t):
    """Cached tensors (packed DeformConv weights, folded BatchNorm affines) a captured graph reads must outlive their
    cache entry: while a GraphedInference capture is collecting, remember them on the graph's owner."""
    if _keepalive is not None and t is not None:
        _keepalive.append(t)
    return t
