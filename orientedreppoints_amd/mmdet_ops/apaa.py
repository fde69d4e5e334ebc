"""Device-side assignment / APAA selection operators (no counterpart module in the reference: there these are Python
loops inside point_assigner.py, max_iou_assigner.py and orientedreppoints_head.py -- see csrc/orp_assign.hip)."""
import ctypes

import torch

from .. import _lib


def point_assign(points, gt_rbboxes, scale=4, pos_num=1):
    """points [N,3] (x,y,stride), gts [K,8] -> gt_inds [N] int64 (0 bg / 1-based gt)."""
    _lib.require_cuda(points, "points")
    p = points.detach().float().contiguous()
    g = gt_rbboxes.detach().float().reshape(-1, 8).contiguous()
    n, k = p.size(0), g.size(0)
    out = torch.empty((n,), dtype=torch.long, device=p.device)
    L = _lib.lib()
    ws = _lib.workspace(p.device, L.orp_point_assign_workspace_bytes(n))
    with torch.cuda.device(p.device):
        rc = L.orp_point_assign(_lib.ptr(p), n, _lib.ptr(g), k, float(scale), int(pos_num), _lib.ptr(out),
                                _lib.ptr(ws), ws.numel(), _lib.stream_of(p))
    _lib.check(rc, "orp_point_assign")
    return out


def max_iou_assign(overlaps_nk, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, gt_max_assign_all=True):
    """overlaps [N,K] point-major -> (gt_inds [N] int64 in {-1,0,1..K}, max_overlaps [N])."""
    _lib.require_cuda(overlaps_nk, "overlaps")
    ov = overlaps_nk.detach().float().contiguous()
    n, k = ov.size(0), ov.size(1)
    if isinstance(neg_iou_thr, (tuple, list)):
        lo, hi = float(neg_iou_thr[0]), float(neg_iou_thr[1])
    else:
        lo, hi = 0.0, float(neg_iou_thr)
    gt_inds = torch.empty((n,), dtype=torch.long, device=ov.device)
    max_ov = torch.empty((n,), dtype=torch.float32, device=ov.device)
    L = _lib.lib()
    ws = _lib.workspace(ov.device, L.orp_max_iou_assign_workspace_bytes(k))
    with torch.cuda.device(ov.device):
        rc = L.orp_max_iou_assign(_lib.ptr(ov), n, k, float(pos_iou_thr), lo, hi, float(min_pos_iou),
                                  int(bool(gt_max_assign_all)), _lib.ptr(gt_inds), _lib.ptr(max_ov), _lib.ptr(ws),
                                  ws.numel(), _lib.stream_of(ov))
    _lib.check(rc, "orp_max_iou_assign")
    return gt_inds, max_ov


def apaa_feature_dissimilarity(feats, strides, pts18, img_index, level_index):
    """feats: list of [B,C,H,W] per level; pts18 [P,18] image-space refined points of the positives -> [P]."""
    p = pts18.detach().float().reshape(-1, 18).contiguous()
    P = p.size(0)
    out = torch.empty((P,), dtype=torch.float32, device=p.device)
    if P == 0:
        return out
    fs = [f.detach().float().contiguous() for f in feats]
    nl = len(fs)
    ptrs = (ctypes.c_void_p * nl)(*[f.data_ptr() for f in fs])
    hs = (ctypes.c_int * nl)(*[f.size(2) for f in fs])
    wsz = (ctypes.c_int * nl)(*[f.size(3) for f in fs])
    st = (ctypes.c_float * nl)(*[float(s) for s in strides])
    ii = img_index.to(device=p.device, dtype=torch.int32).contiguous()
    li = level_index.to(device=p.device, dtype=torch.int32).contiguous()
    with torch.cuda.device(p.device):
        rc = _lib.lib().orp_apaa_feature_dissimilarity(ptrs, hs, wsz, st, nl, fs[0].size(1), _lib.ptr(p), _lib.ptr(ii),
                                                       _lib.ptr(li), P, _lib.ptr(out), _lib.stream_of(p))
    _lib.check(rc, "orp_apaa_feature_dissimilarity")
    return out


def apaa_select(quality, pos_gt_inds, pos_level, num_gt, num_level, per_level_topk=6, top_ratio=0.4):
    """-> keep [P] bool: the positives that survive the per-gt quality selection."""
    q = quality.detach().float().contiguous()
    P = q.size(0)
    keep = torch.zeros((P,), dtype=torch.uint8, device=q.device)
    if P == 0:
        return keep.bool()
    g = pos_gt_inds.to(torch.long).contiguous()
    lv = pos_level.to(torch.int32).contiguous()
    with torch.cuda.device(q.device):
        rc = _lib.lib().orp_apaa_select(_lib.ptr(q), _lib.ptr(g), _lib.ptr(lv), P, int(num_gt), int(num_level),
                                        int(per_level_topk), float(top_ratio), _lib.ptr(keep), _lib.stream_of(q))
    _lib.check(rc, "orp_apaa_select")
    return keep.bool()
