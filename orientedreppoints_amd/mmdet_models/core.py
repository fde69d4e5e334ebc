"""Small host utilities of mmdet.core that the dense head uses, restated for the MI355X build:
  multi_apply / unmap                mmdet/core/utils/misc.py:21-37
  PointGenerator                     mmdet/core/anchor/point_generator.py:4-34
  levels_to_images                   mmdet/core/anchor/anchor_target.py:172-186
  multiclass_rnms                    mmdet/core/post_processing/bbox_nms.py:93-182
  rbbox2result                       mmdet/core/bbox/transforms.py:356-375
"""
from functools import partial

import numpy as np
import torch

from .. import _lib
from ..mmdet_ops import nms_wrapper


def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    map_results = map(pfunc, *args)
    return tuple(map(list, zip(*map_results)))


def unmap(data, count, inds, fill=0):
    """Unmap a subset of items (data) back to the original set of items (of size count)."""
    if data.dim() == 1:
        ret = data.new_full((count, ), fill)
        ret[inds] = data
    else:
        new_size = (count, ) + data.size()[1:]
        ret = data.new_full(new_size, fill)
        ret[inds, :] = data
    return ret


class PointGenerator(object):

    def _meshgrid(self, x, y, row_major=True):
        xx = x.repeat(len(y))
        yy = y.view(-1, 1).repeat(1, len(x)).view(-1)
        return (xx, yy) if row_major else (yy, xx)

    def grid_points(self, featmap_size, stride=16, device='cuda'):
        """[H*W, 3] = (x, y, stride) (point_generator.py:14-22).  The grid depends only on (H, W, stride, device): it is
        built once and the same read-only tensor is handed out afterwards (8 small launches per level otherwise)."""
        feat_h, feat_w = featmap_size
        key = (int(feat_h), int(feat_w), float(stride), str(device))
        cache = self.__dict__.setdefault('_grid_cache', {})
        if key not in cache:
            if len(cache) >= 64:                               # value-keyed; drop the oldest entry only
                cache.pop(next(iter(cache)))
            cache[key] = self._grid_points(featmap_size, stride, device)
        return _lib.keep_for_graph(cache[key])

    def _grid_points(self, featmap_size, stride, device):
        feat_h, feat_w = featmap_size
        shift_x = torch.arange(0., feat_w, device=device) * stride
        shift_y = torch.arange(0., feat_h, device=device) * stride
        shift_xx, shift_yy = self._meshgrid(shift_x, shift_y)
        stride = shift_x.new_full((shift_xx.shape[0], ), stride)
        return torch.stack([shift_xx, shift_yy, stride], dim=-1)

    def valid_flags(self, featmap_size, valid_size, device='cuda'):
        feat_h, feat_w = featmap_size
        valid_h, valid_w = valid_size
        assert valid_h <= feat_h and valid_w <= feat_w
        valid_x = torch.zeros(feat_w, dtype=torch.bool, device=device)
        valid_y = torch.zeros(feat_h, dtype=torch.bool, device=device)
        valid_x[:valid_w] = 1
        valid_y[:valid_h] = 1
        valid_xx, valid_yy = self._meshgrid(valid_x, valid_y)
        return valid_xx & valid_yy


def levels_to_images(mlvl_tensor, flatten=False):
    """[lvl][B,C,H,W] -> [img][sum(HW), C]  (or [sum(HW)*C/...] flattened per the reference's flag)."""
    batch_size = mlvl_tensor[0].size(0)
    batch_list = [[] for _ in range(batch_size)]
    if flatten:
        channels = mlvl_tensor[0].size(-1)
    else:
        channels = mlvl_tensor[0].size(1)
    for t in mlvl_tensor:
        if not flatten:
            t = t.permute(0, 2, 3, 1)
        t = t.view(batch_size, -1, channels).contiguous()
        for img in range(batch_size):
            batch_list[img].append(t[img])
    return [torch.cat(item, 0) for item in batch_list]


def multiclass_rnms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None,
                    multi_reppoints=None):
    """Reference semantics (bbox_nms.py:93-182), including the class-offset trick on fp32 coordinates: every
    (point, class) pair above score_thr is a detection; one rotated NMS over `coords + label * (max + 1)`."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 8:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 8)[:, 1:]
    else:
        bboxes = multi_bboxes[:, None].expand(-1, num_classes, 8)
    if multi_reppoints is not None:
        reppoints = multi_reppoints[:, None].expand(-1, num_classes, multi_reppoints.size(-1))
    scores = multi_scores[:, 1:]
    valid_mask = scores > score_thr
    bboxes = bboxes[valid_mask]
    if multi_reppoints is not None:
        reppoints = reppoints[valid_mask]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid_mask]
    labels = valid_mask.nonzero()[:, 1]
    if bboxes.numel() == 0:
        if multi_reppoints is None:
            bboxes = multi_bboxes.new_zeros((0, 9))
        else:
            bboxes = multi_bboxes.new_zeros((0, reppoints.size(-1) + 9))
        labels = multi_bboxes.new_zeros((0, ), dtype=torch.long)
        return bboxes, labels
    max_coordinate = bboxes.max()
    offsets = labels.to(bboxes) * (max_coordinate + 1)
    bboxes_for_nms = bboxes + offsets[:, None]
    nms_cfg_ = dict(nms_cfg)
    nms_type = nms_cfg_.pop('type', 'rnms')
    nms_op = getattr(nms_wrapper, nms_type)
    dets, keep = nms_op(torch.cat([bboxes_for_nms, scores[:, None]], 1), **nms_cfg_)
    bboxes = bboxes[keep]
    if multi_reppoints is not None:
        reppoints = reppoints[keep]
        bboxes = torch.cat([reppoints, bboxes], dim=1)
    scores = dets[:, -1]
    labels = labels[keep]
    if keep.size(0) > max_num:
        _, inds = scores.sort(descending=True)
        inds = inds[:max_num]
        bboxes = bboxes[inds]
        scores = scores[inds]
        labels = labels[inds]
    return torch.cat([bboxes, scores[:, None]], 1), labels


def multiclass_rnms_static(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num, multi_reppoints, capacity=8192):
    """`multiclass_rnms` with the reference's exact semantics (bbox_nms.py:93-182: same detections, same order, the
    class-offset trick on fp32 coordinates) but with STATIC shapes and no host synchronisation: boolean-mask indexing /
    nonzero (one blocking D2H each) become a cumsum + scatter into a fixed-capacity buffer, the NMS takes its box
    count from device memory (`orp_rnms_batched`, one segment [0, n)), the `> max_num` re-sort is a top-k selected on
    device.  Everything between the head's convolutions and the final result copy is stream-ordered and
    hipGraph-capturable.

    Returns a packed float32 tensor [max_num + 1, W + 2]: rows = [reppoints | 8 corners | score | label]; the last row
    holds (count, overflow).  overflow = more than `capacity` (point, class) pairs passed score_thr -> the caller must
    fall back to the dynamic path."""
    M0 = multi_scores.size(0)
    C = multi_scores.size(1) - 1
    dev = multi_scores.device
    assert multi_bboxes.shape[1] == 8, 'static path: class-agnostic boxes [M0, 8]'
    flat = multi_scores[:, 1:].reshape(-1)
    numel = flat.numel()
    cap = int(min(capacity, numel))
    valid = flat > score_thr
    total = valid.sum()
    pos = torch.cumsum(valid, 0) - 1
    slot = torch.where(valid & (pos < cap), pos, torch.full_like(pos, cap))
    idx = torch.zeros(cap + 1, dtype=torch.long, device=dev).scatter_(0, slot, torch.arange(numel, device=dev))[:cap]
    n = torch.clamp(total, max=cap)
    ar = torch.arange(cap, device=dev)
    live = ar < n
    pt, lab = idx // C, idx % C
    boxes = multi_bboxes[pt]
    neg_inf = torch.full((), float('-inf'), device=dev)
    sc = torch.where(live, flat[idx], neg_inf)
    max_coordinate = torch.where(live[:, None], boxes, neg_inf).max()
    offsets = lab.to(boxes) * (max_coordinate + 1)
    dets = torch.cat([boxes + offsets[:, None], sc[:, None]], 1)
    nms_cfg_ = dict(nms_cfg)
    nms_type = nms_cfg_.pop('type', 'rnms')
    assert nms_type == 'rnms', 'static path implements the rnms configuration of the DOTA configs'
    seg = torch.stack([torch.zeros_like(n), n]).to(torch.int32)
    keep, num = nms_wrapper.rnms_batched_device(dets, seg, cap, nms_cfg_.get('iou_thr', 0.4))
    kn = num[0].to(torch.long)
    livek = ar < kn
    sel = torch.where(livek, keep, torch.zeros_like(keep))
    k_scores = torch.where(livek, sc[sel], neg_inf)
    m = int(min(max_num, cap)) if max_num > 0 else cap
    # reference: keep order (ascending index) unless more than max_num survive, then score-descending top max_num
    _, top_i = k_scores.topk(m)
    pick = torch.where(kn > m, top_i, ar[:m])
    rows = sel[pick]
    count = torch.clamp(kn, max=m)
    out_live = (ar[:m] < count)
    body = torch.cat([multi_reppoints[pt[rows]], boxes[rows], sc[rows][:, None], lab[rows].to(boxes)[:, None]], 1)
    body = torch.where(out_live[:, None], body, torch.zeros_like(body))
    tail = torch.zeros(1, body.size(1), dtype=body.dtype, device=dev)
    tail[0, 0] = count.to(body.dtype)
    tail[0, 1] = (total > cap).to(body.dtype)
    return torch.cat([body, tail], 0)


def fused_postprocess(cls_scores, points_preds, strides, cfg, num_points=9):
    """get_bboxes_single + multiclass_rnms + packing for ONE image on the fused HIP kernels (csrc/orp_postproc.hip):
    cls_scores[l] [C,H,W] logits, points_preds[l] [2*num_points,H,W] refine offsets ((y,x)-interleaved, grid units).
    Same detections in the same order as the tensor-op path (`multiclass_rnms_static`); torch keeps the numerically
    sensitive pieces (sigmoid, class max, top-k), everything else is three kernels around min-area-rect and the NMS.
    Returns the packed [max_per_img + 1, 28] tensor of `rbbox2result_packed`."""
    import ctypes
    from ..mmdet_ops.minarea_rect import minaerarect_decode
    assert num_points == 9
    L = _lib.lib()
    dev = cls_scores[0].device
    C = cls_scores[0].size(0)
    sizes = [int(c.size(1)) * int(c.size(2)) for c in cls_scores]
    widths = [int(c.size(2)) for c in cls_scores]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N = int(offs[-1])
    # (fp32 from here on: the kernels below take float pointers -- a half / bfloat16 model's head outputs are widened once)
    logits = torch.cat([c.reshape(C, -1) for c in cls_scores], 1).float()            # [C, N]
    pts_all = torch.cat([p.reshape(2 * num_points, -1) for p in points_preds], 1).float().contiguous()   # [18, N]
    sig = logits.sigmoid()
    nms_pre = cfg.get('nms_pre', -1)
    if 0 < nms_pre <= 4096 and any(n_l > nms_pre for n_l in sizes) and len(sizes) <= 8 and max(sizes) <= 40960:
        cand = select_candidates(sig, offs, nms_pre)
    else:
        cand = []
        max_scores = None
        for l, n_l in enumerate(sizes):
            if nms_pre > 0 and n_l > nms_pre:
                if max_scores is None:
                    max_scores = sig.max(dim=0)[0]
                _, topk_inds = max_scores[int(offs[l]):int(offs[l + 1])].topk(nms_pre)
                cand.append(topk_inds + int(offs[l]))
            else:
                cand.append(_arange_cached(int(offs[l]), int(offs[l + 1]), dev))
        cand = torch.cat(cand)
    m0 = cand.numel()
    f32 = dict(dtype=torch.float32, device=dev)
    pts_xy = torch.empty((m0, 18), **f32)
    centers = torch.empty((m0, 2), **f32)
    strd = torch.empty((m0,), **f32)
    rep = torch.empty((m0, 18), **f32)
    lo = (ctypes.c_int * len(sizes))(*[int(o) for o in offs[:-1]])
    lw = (ctypes.c_int * len(sizes))(*widths)
    ls = (ctypes.c_float * len(sizes))(*[float(s) for s in strides])
    st = _lib.stream_of(sig)
    with torch.cuda.device(dev):
        _lib.check(L.orp_pp_gather(_lib.ptr(pts_all), _lib.ptr(cand), m0, N, lo, lw, ls, len(sizes), _lib.ptr(pts_xy),
                                   _lib.ptr(centers), _lib.ptr(strd), _lib.ptr(rep), st), "orp_pp_gather")
    boxes = minaerarect_decode(pts_xy, centers, strd)                                # [m0, 8]
    cap = int(min(cfg.get('static_capacity', 8192), m0 * C))
    max_num = int(cfg.max_per_img)
    m = int(min(max_num, cap)) if max_num > 0 else cap
    dets = torch.empty((cap, 9), **f32)
    sel_cand = torch.empty((cap,), dtype=torch.int32, device=dev)
    sel_label = torch.empty((cap,), dtype=torch.int32, device=dev)
    seg = torch.empty((2,), dtype=torch.int32, device=dev)
    total = torch.empty((1,), dtype=torch.int32, device=dev)
    nms_cfg_ = dict(cfg.nms)
    assert nms_cfg_.pop('type', 'rnms') == 'rnms'
    with torch.cuda.device(dev):
        scratch = torch.empty((L.orp_pp_compact_scratch_bytes(m0),), dtype=torch.uint8, device=dev)
        _lib.check(L.orp_pp_compact(_lib.ptr(sig), _lib.ptr(cand), m0, N, C, _lib.ptr(boxes), float(cfg.score_thr), cap,
                                    _lib.ptr(dets), _lib.ptr(sel_cand), _lib.ptr(sel_label), _lib.ptr(seg),
                                    _lib.ptr(total), _lib.ptr(scratch), scratch.numel(), st), "orp_pp_compact")
    keep, num = nms_wrapper.rnms_batched_device(dets, seg, cap, nms_cfg_.get('iou_thr', 0.4))
    packed = torch.empty((m + 1, 28), **f32)
    with torch.cuda.device(dev):
        _lib.check(L.orp_pp_pack(_lib.ptr(keep), _lib.ptr(num), _lib.ptr(dets), _lib.ptr(sel_cand), _lib.ptr(sel_label),
                                 _lib.ptr(boxes), _lib.ptr(rep), _lib.ptr(total), cap, m, _lib.ptr(packed), st),
                   "orp_pp_pack")
    return packed


def select_candidates(sig, offs, nms_pre):
    """`scores.max(dim=1)` + per-level `topk(nms_pre)` of get_bboxes_single (head :730-737) as one radix select + counting
    rank (`orp_pp_select`): sig [C, N] sigmoid scores of all levels, offs = first point of each level (+ N).  Returns the
    int64 candidate indices, level-major, descending class-maximum score inside a selected level."""
    import ctypes
    L = _lib.lib()
    C, N = int(sig.size(0)), int(sig.size(1))
    nlev = len(offs) - 1
    sizes = [int(offs[i + 1] - offs[i]) for i in range(nlev)]
    m0 = sum(min(n_l, nms_pre) for n_l in sizes)
    cand = torch.empty((m0,), dtype=torch.int64, device=sig.device)
    lo = (ctypes.c_int * nlev)(*[int(o) for o in offs[:-1]])
    nbytes = L.orp_pp_select_scratch_bytes(N, nms_pre, nlev)
    with torch.cuda.device(sig.device):
        scratch = torch.empty((nbytes,), dtype=torch.uint8, device=sig.device)
        _lib.check(L.orp_pp_select(_lib.ptr(sig), C, N, lo, nlev, int(nms_pre), _lib.ptr(cand), _lib.ptr(scratch), nbytes,
                                   _lib.stream_of(sig)), "orp_pp_select")
    return cand


_arange_cache = {}


def _arange_cached(a, b, dev):
    """Index range of one FPN level (value-keyed, so a hit can never be a different range); a captured graph that
    reads it keeps it alive itself (`keep_for_graph`), eviction drops the oldest entry only."""
    key = (a, b, dev)
    t = _arange_cache.get(key)
    if t is None:
        t = torch.arange(a, b, device=dev)
        if len(_arange_cache) >= 256:
            _arange_cache.pop(next(iter(_arange_cache)))
        _arange_cache[key] = t
    return _lib.keep_for_graph(t)


def rbbox2result_packed(packed, num_classes):
    """One D2H copy of the packed static result -> the reference's per-class list (rbbox2result).  Returns None when
    the capacity overflowed (caller falls back to the dynamic path)."""
    host = packed.cpu().numpy()
    count, overflow = int(host[-1, 0]), bool(host[-1, 1])
    if overflow:
        return None
    if count == 0:
        return [np.zeros((0, 9), dtype=np.float32) for _ in range(num_classes - 1)]
    bboxes = host[:count, :-1]
    labels = host[:count, -1].astype(np.int64)
    return [bboxes[labels == i, :] for i in range(num_classes - 1)]


def rbbox2result(bboxes, labels, num_classes):
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 9), dtype=np.float32) for _ in range(num_classes - 1)]
    bboxes = bboxes.cpu().numpy()
    labels = labels.cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes - 1)]
