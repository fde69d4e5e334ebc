"""Batched glue operators of the training path (include/orp_hip.h `orp_pointset_target`, `orp_points_from_offsets`,
`orp_gather_levels`, `orp_outline_samples`; csrc/orp_train.hip).  No counterpart module in the reference: there this work
is per-image / per-level Python loops of small tensor operations inside pointset_target.py and
orientedreppoints_head.py (loss, offset_to_pts, sampling_points)."""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .. import _lib


class _LevelDesc(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("height", ctypes.c_int), ("width", ctypes.c_int),
                ("stride", ctypes.c_float)]


def _level_table(levels, strides, grads=None):
    """levels: list of contiguous fp32 [B,C,H,W] tensors (same B, C)."""
    n = len(levels)
    tab = (_LevelDesc * n)()
    B, C = levels[0].size(0), levels[0].size(1)
    for i, t in enumerate(levels):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.is_contiguous() and t.size(0) == B and
                t.size(1) == C):
            raise ValueError("level tensors must be contiguous fp32 CUDA [B,C,H,W] with equal B and C")
        g = grads[i] if grads is not None else None
        tab[i] = _LevelDesc(t.data_ptr(), g.data_ptr() if g is not None else None, t.size(2), t.size(3), float(strides[i]))
    return tab, B, C, sum(int(t.size(2) * t.size(3)) for t in levels)


def points_from_offsets(levels, strides, mode=0):
    """[lvl][B,18,H,W] offset maps -> [B, N, 18] point sets of every location (no autograd).  mode 0: offset_to_pts
    (image-space (x, y) pairs); mode 1: the refine-stage proposals of loss() (centre + offset * stride, no (y,x) swap)."""
    lv = [t.detach().float().contiguous() for t in levels]
    tab, B, C, N = _level_table(lv, strides)
    out = torch.empty((B, N, C), dtype=torch.float32, device=lv[0].device)
    with torch.cuda.device(out.device):
        rc = _lib.lib().orp_points_from_offsets(tab, len(lv), B, C, int(mode), _lib.ptr(out), _lib.stream_of(out))
    _lib.check(rc, "orp_points_from_offsets")
    return out


class _GatherLevels(Function):

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: fp32 inputs, autocast off inside
    def forward(ctx, index, strides, mode, *levels):
        lv = [t.detach().float().contiguous() for t in levels]
        tab, B, C, N = _level_table(lv, strides)
        P = index.numel()
        out = torch.empty((P, C), dtype=torch.float32, device=lv[0].device)
        idx = index.to(torch.long).contiguous()
        if P:
            with torch.cuda.device(out.device):
                rc = _lib.lib().orp_gather_levels(tab, len(lv), B, C, _lib.ptr(idx), P, int(mode), _lib.ptr(out),
                                                  _lib.stream_of(out))
            _lib.check(rc, "orp_gather_levels")
        ctx.save_for_backward(idx)
        ctx.meta = (tuple(float(s) for s in strides), int(mode), [tuple(t.shape) for t in levels], lv[0].device)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        strides, mode, shapes, dev = ctx.meta
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in shapes]      # zero-filled by the C entry
        tab, B, C, N = _level_table(grads, strides, grads)
        go = grad_out.detach().float().contiguous()
        with torch.cuda.device(dev):
            rc = _lib.lib().orp_gather_levels_backward(tab, len(grads), B, C, _lib.ptr(idx), idx.numel(), mode,
                                                       _lib.ptr(go), _lib.stream_of(go))
        _lib.check(rc, "orp_gather_levels_backward")
        return (None, None, None) + tuple(grads)


def gather_levels(levels, strides, index, mode=0):
    """Rows of [lvl][B,C,H,W] tensors at the locations index[p] = b * N + i -> [P, C] (differentiable w.r.t. the
    levels; the selected locations must be distinct).  mode 1: as image-space point sets (offset_to_pts)."""
    return _GatherLevels.apply(index, strides, mode, *levels)


def pointset_target(gt_inds, valid, gt_boxes, gt_labels, gt_offset, pos_weight=-1.0, proposals=None, want_counts=True):
    """All images' assignment results -> their full-N targets in one launch.  gt_inds [B,N] int64, valid [B,N] bool or
    None, gt_boxes [K,8] / gt_labels [K] (all images concatenated), gt_offset [B+1] int32 on the device.
    Returns dict(labels, label_weights, rbbox_gt, proposal_weights, gt_inds, pos_proposals?, counts [B,2] int32?)."""
    B, N = gt_inds.shape
    dev = gt_inds.device
    gi = gt_inds.to(torch.long).contiguous()
    v = valid.to(torch.uint8).contiguous() if valid is not None else None
    gb = gt_boxes.detach().float().reshape(-1, 8).contiguous()
    gl = gt_labels.to(torch.long).contiguous() if gt_labels is not None else None
    go = gt_offset.to(device=dev, dtype=torch.int32).contiguous()
    out = dict(labels=torch.empty((B, N), dtype=torch.long, device=dev),
               label_weights=torch.empty((B, N), dtype=torch.float32, device=dev),
               rbbox_gt=torch.empty((B, N, 8), dtype=torch.float32, device=dev),
               proposal_weights=torch.empty((B, N), dtype=torch.float32, device=dev),
               gt_inds=torch.empty((B, N), dtype=torch.long, device=dev))
    pp, D, prop = None, 0, None
    if proposals is not None:
        prop = proposals.detach().float().contiguous()
        D = prop.size(-1)
        pp = out['pos_proposals'] = torch.empty_like(prop)
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev) if want_counts else None
    if counts is not None:
        out['counts'] = counts
    with torch.cuda.device(dev):
        rc = _lib.lib().orp_pointset_target(_lib.ptr(gi), _lib.ptr(v), B, N, _lib.ptr(gb), _lib.ptr(gl), _lib.ptr(go),
                                            _lib.ptr(prop), D, float(pos_weight), _lib.ptr(out['labels']),
                                            _lib.ptr(out['label_weights']), _lib.ptr(out['rbbox_gt']), _lib.ptr(pp),
                                            _lib.ptr(out['proposal_weights']), _lib.ptr(out['gt_inds']), _lib.ptr(counts),
                                            _lib.stream_of(gi))
    _lib.check(rc, "orp_pointset_target")
    return out


class _SegmentGIoULoss(Function):
    """loss[s] = loss_weight * sum_{i in s} w_i (1 - GIoU_i) / max(denom[s], 1) for the rows' segments s (GIoULoss with
    reduction 'mean' applied per segment, iou_loss.py:69-129).  As in the reference the gradient comes out of the forward
    kernel: d/d pred_i = -grad_i w_i / denom[seg_i] * loss_weight, rows with any component > 1 replaced by 1e-6 first
    (iou_loss.py:87-89), times the incoming gradient of the row's segment (the reference ignores it; it is 1 in the
    reference's training loop, and under fp16 loss scaling it must not be dropped).  convex_giou + two launches (rows,
    fixed-order segment sum)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: fp32 inputs, autocast off inside
    def forward(ctx, pred, target, weight, seg, nseg, denom, loss_weight):
        from .iou_wrapper import convex_giou
        P = pred.size(0)
        dev = pred.device
        loss = torch.zeros((nseg,), dtype=torch.float32, device=dev)
        if P == 0:                                                       # no row at all: zero loss, empty gradient
            ctx.save_for_backward(torch.zeros_like(pred), torch.zeros((0,), dtype=torch.long, device=dev))
            return loss
        gious, grad = convex_giou(pred, target)
        gious = gious.float().contiguous()
        grad = grad.float().reshape(P, 18).contiguous()
        w = weight.detach().float().contiguous()
        sg = seg.to(torch.long).contiguous()
        d = denom.detach().float().reshape(nseg).contiguous()
        contrib = torch.empty((P,), dtype=torch.float32, device=dev)
        gsave = torch.empty((P, 18), dtype=torch.float32, device=dev)
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = _lib.stream_of(gious)
            _lib.check(L.orp_giou_rows(_lib.ptr(gious), _lib.ptr(grad), _lib.ptr(w), _lib.ptr(sg), _lib.ptr(d), P,
                                       float(loss_weight), _lib.ptr(contrib), _lib.ptr(gsave), st), "orp_giou_rows")
            _lib.check(L.orp_segment_finish(_lib.ptr(contrib), None, _lib.ptr(sg), P, int(nseg), _lib.ptr(d),
                                            float(loss_weight), 0, _lib.ptr(loss), None, st), "orp_segment_finish")
        ctx.save_for_backward(gsave.reshape(pred.shape), sg)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out=None):
        g, sg = ctx.saved_tensors
        if grad_out is not None and g.numel():
            g = g * grad_out.to(g.dtype).reshape(-1)[sg].reshape((-1,) + (1,) * (g.dim() - 1))
        return g, None, None, None, None, None, None


def segment_giou_loss(pred, target, weight, seg, nseg, denom, loss_weight):
    return _SegmentGIoULoss.apply(pred, target, weight, seg, nseg, denom, loss_weight)


class _SegmentBorderLoss(Function):
    """SpatialBorderLoss per segment (spatial_border_loss.py:8-92): for the points of rows with weight > 0 that lie
    outside their gt quad, 0.2 * distance to the quad centre, summed, divided by the number of such points and by
    denom[s] + 1e-6.  Two launches forward (rows, fixed-order segment sums), the stored direction field backward."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, pts, gt, weight, seg, nseg, denom, loss_weight):
        P = pts.size(0)
        dev = pts.device
        loss = torch.zeros((nseg,), dtype=torch.float32, device=dev)
        if P == 0:
            ctx.empty = True
            ctx.shape = tuple(pts.shape)
            return loss
        ctx.empty = False
        p = pts.detach().float().reshape(P, 18).contiguous()
        g = gt.detach().float().reshape(P, 8).contiguous()
        w = weight.detach().float().contiguous()
        sg = seg.to(torch.long).contiguous()
        d = denom.detach().float().reshape(nseg).contiguous()
        row_sum = torch.empty((P,), dtype=torch.float32, device=dev)
        row_cnt = torch.empty((P,), dtype=torch.float32, device=dev)
        gdir = torch.empty((P, 18), dtype=torch.float32, device=dev)
        scale = torch.empty((nseg,), dtype=torch.float32, device=dev)
        L = _lib.lib()
        with torch.cuda.device(dev):
            st = _lib.stream_of(p)
            _lib.check(L.orp_border_rows(_lib.ptr(p), _lib.ptr(g), _lib.ptr(w), P, _lib.ptr(row_sum), _lib.ptr(row_cnt),
                                         _lib.ptr(gdir), st), "orp_border_rows")
            _lib.check(L.orp_segment_finish(_lib.ptr(row_sum), _lib.ptr(row_cnt), _lib.ptr(sg), P, int(nseg), _lib.ptr(d),
                                            float(loss_weight), 1, _lib.ptr(loss), _lib.ptr(scale), st), "orp_segment_finish")
        ctx.save_for_backward(gdir, scale, sg)
        ctx.shape = tuple(pts.shape)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        if ctx.empty:
            return grad_out.new_zeros(ctx.shape), None, None, None, None, None, None
        gdir, scale, sg = ctx.saved_tensors
        return (gdir * (grad_out.float() * scale)[sg][:, None]).reshape(ctx.shape), None, None, None, None, None, None


def segment_border_loss(pts, gt, weight, seg, nseg, denom, loss_weight):
    return _SegmentBorderLoss.apply(pts, gt, weight, seg, nseg, denom, loss_weight)


_ratios = {}


def outline_samples(corners, n):
    """[P,8] quads -> [P, 4n, 2]: n points at torch.linspace(0, 1, n) ratios on each edge 1->2->3->4->1 (no autograd)."""
    c = corners.detach().float().reshape(-1, 8).contiguous()
    P = c.size(0)
    out = torch.empty((P, 4 * n, 2), dtype=torch.float32, device=c.device)
    if P:
        key = (int(n), c.device)
        r = _ratios.get(key)
        if r is None:
            r = _ratios[key] = torch.linspace(0, 1, int(n), device=c.device)     # the reference's own ratios, built once
        with torch.cuda.device(c.device):
            rc = _lib.lib().orp_outline_samples(_lib.ptr(c), P, int(n), _lib.ptr(r), _lib.ptr(out), _lib.stream_of(c))
        _lib.check(rc, "orp_outline_samples")
    return out
