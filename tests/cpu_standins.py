"""TEST INFRASTRUCTURE (never imported by the product): CPU stand-ins for the HIP operators, backed by the oracle
(oracle/orp_oracle.py) and plain tensor code, so that the real detector -- its own modules, loss() and autograd graph -- can
take a training step on the CPU.  Used by the 2-rank gloo test of the gradient exchange (tests/test_distributed_detector.py):
what is under test there is the host logic around the operators (the reducer against the detector's ragged autograd graph),
not the operators, which the -m gpu tests check against the same oracle."""
import contextlib
import importlib

import numpy as np
import torch
from torch.autograd import Function

from oracle import orp_oracle as O


def _np(t):
    return t.detach().cpu().numpy()


class _CpuDeformConv(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
        s = stride[0] if isinstance(stride, (tuple, list)) else stride
        p = padding[0] if isinstance(padding, (tuple, list)) else padding
        d = dilation[0] if isinstance(dilation, (tuple, list)) else dilation
        assert groups == 1 and deformable_groups == 1
        ctx.geo = (s, p, d)
        ctx.save_for_backward(input, offset, weight)
        return torch.from_numpy(O.dcn_forward(_np(input), _np(offset), _np(weight), stride=s, pad=p, dil=d))

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        s, p, d = ctx.geo
        gi, goff, gw = O.dcn_backward(_np(input), _np(offset), _np(weight), _np(grad_output.contiguous()), stride=s, pad=p, dil=d)
        return (torch.from_numpy(gi), torch.from_numpy(goff), torch.from_numpy(gw), None, None, None, None, None, None)


class _CpuModulatedDeformConv(Function):
    """ModulatedDeformConvFunction on the CPU through oracle.dcn_v2_forward / dcn_v2_backward (groups = 1)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
        s = stride[0] if isinstance(stride, (tuple, list)) else stride
        p = padding[0] if isinstance(padding, (tuple, list)) else padding
        d = dilation[0] if isinstance(dilation, (tuple, list)) else dilation
        assert groups == 1
        ctx.geo = (s, p, d, deformable_groups)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(input, offset, mask, weight)
        return torch.from_numpy(O.dcn_v2_forward(_np(input), _np(offset), _np(mask), _np(weight),
                                                 _np(bias) if bias is not None else None, s, p, d, deformable_groups))

    @staticmethod
    def backward(ctx, grad_output):
        input, offset, mask, weight = ctx.saved_tensors
        s, p, d, dg = ctx.geo
        gi, goff, gm, gw, gb = O.dcn_v2_backward(_np(input), _np(offset), _np(mask), _np(weight),
                                                 _np(grad_output.contiguous()), s, p, d, dg)
        return (torch.from_numpy(gi), torch.from_numpy(goff), torch.from_numpy(gm), torch.from_numpy(gw),
                torch.from_numpy(gb) if ctx.has_bias else None, None, None, None, None, None)


def _focal(logits, targets, gamma, alpha):
    """sigmoid focal loss per element (sigmoid_focal_loss_cuda.cu:23-97), labels 1..C, 0 = background -- differentiable."""
    C = logits.size(1)
    t = torch.nn.functional.one_hot(targets.clamp(min=0), C + 1)[:, 1:].to(logits.dtype)
    p = torch.sigmoid(logits)
    pos = -alpha * (1 - p) ** gamma * torch.log(p.clamp(min=1.1754943508222875e-38))
    neg = -(1 - alpha) * p ** gamma * (-logits * (logits >= 0).to(logits.dtype)
                                        - torch.log1p(torch.exp(logits - 2 * logits * (logits >= 0).to(logits.dtype))))
    return t * pos + (1 - t) * neg


def _convex_giou(pred, target):
    if pred.numel() == 0:
        return torch.zeros((0,)), torch.zeros((0, 18))
    out = torch.from_numpy(O.convex_giou(_np(pred).reshape(-1, 18), _np(target).reshape(-1, 8)))
    return out[:, -1], out[:, :-1]


def _convex_iou(pred, target):
    if pred.numel() == 0 or target.numel() == 0:
        return torch.zeros((pred.size(0), target.size(0)))
    return torch.from_numpy(O.convex_iou(_np(pred).reshape(-1, 18), _np(target).reshape(-1, 8)).astype(np.float32))


def _point_assign(points, gts, scale=4, pos_num=1):
    return torch.from_numpy(O.point_assign(_np(points), _np(gts), scale, pos_num))


def _max_iou_assign(ov, pos_thr, neg_thr, min_pos=0.0, assign_all=True):
    gi, mo = O.max_iou_assign(_np(ov), pos_thr, tuple(neg_thr) if isinstance(neg_thr, (tuple, list)) else neg_thr, min_pos, assign_all)
    return torch.from_numpy(gi), torch.from_numpy(mo)


def _feature_dissimilarity(feats, strides, pts18, img_index, level_index):
    P = pts18.size(0)
    out = np.zeros((P,), np.float32)
    p, ii, li = _np(pts18), _np(img_index), _np(level_index)
    for k in range(P):
        f = O.sample_points(_np(feats[int(li[k])][int(ii[k])]), float(strides[int(li[k])]), p[k:k + 1])
        out[k] = O.feature_dissimilarity(f)[0]
    return torch.from_numpy(out)


def _apaa_select(q, pos_gt, pos_lvl, num_gt, num_level, k=6, ratio=0.4):
    if q.numel() == 0:
        return torch.zeros((0,), dtype=torch.bool)
    return torch.from_numpy(O.apaa_select(_np(q), _np(pos_gt), _np(pos_lvl), num_gt, num_level, k, ratio)).bool()


def _centres(levels, strides):
    out = []
    for t, s in zip(levels, strides):
        h, w = t.shape[2:]
        ys, xs = torch.meshgrid(torch.arange(h) * float(s), torch.arange(w) * float(s), indexing="ij")
        out.append(torch.stack([xs.reshape(-1), ys.reshape(-1)], 1))
    return out


def _points_all(levels, strides, mode):
    B = levels[0].size(0)
    rows = []
    for t, s, c in zip(levels, strides, _centres(levels, strides)):
        v = t.permute(0, 2, 3, 1).reshape(B, -1, t.size(1))
        if mode == 0:
            v = v.reshape(B, -1, t.size(1) // 2, 2).flip(-1).reshape(B, -1, t.size(1))
        rows.append(v * s + c.repeat(1, t.size(1) // 2))
    return torch.cat(rows, 1)


def _points_from_offsets(levels, strides, mode=0):
    return _points_all([t.detach() for t in levels], strides, mode)


def _gather_levels(levels, strides, index, mode=0):
    B, C = levels[0].size(0), levels[0].size(1)
    flat = _points_all(levels, strides, 0) if mode == 1 else \
        torch.cat([t.permute(0, 2, 3, 1).reshape(B, -1, C) for t in levels], 1)
    return flat.reshape(-1, C)[index]


def _pointset_target(gt_inds, valid, gt_boxes, gt_labels, gt_offset, pos_weight=-1.0, proposals=None, want_counts=True):
    B, N = gt_inds.shape
    g = gt_inds.clone() if valid is None else torch.where(valid, gt_inds, torch.zeros_like(gt_inds))
    pos, neg = g > 0, (g == 0) if valid is None else ((g == 0) & valid)
    k = (gt_offset.long()[:-1, None] + (g - 1).clamp(min=0))
    lab = (gt_labels[k] if gt_labels is not None else torch.ones_like(g)) if gt_boxes.size(0) else torch.zeros_like(g)
    box = gt_boxes[k] if gt_boxes.size(0) else torch.zeros((B, N, 8))
    out = dict(labels=torch.where(pos, lab, torch.zeros_like(lab)),
               label_weights=torch.where(pos, torch.full((B, N), 1.0 if pos_weight <= 0 else pos_weight),
                                         torch.where(neg, torch.ones((B, N)), torch.zeros((B, N)))),
               rbbox_gt=torch.where(pos[..., None], box, torch.zeros_like(box)), proposal_weights=pos.float(), gt_inds=g)
    if proposals is not None:
        out['pos_proposals'] = torch.where(pos[..., None], proposals, torch.zeros_like(proposals))
    if want_counts:
        out['counts'] = torch.stack([pos.sum(1), neg.sum(1)], 1).to(torch.int32)
    return out


def _outline_samples(corners, n):
    q = corners.detach().reshape(-1, 4, 2)
    r = torch.linspace(0, 1, n).view(1, 1, n, 1)
    return (r * torch.roll(q, -1, 1).unsqueeze(2) + (1 - r) * q.unsqueeze(2)).reshape(q.size(0), 4 * n, 2)


def _chamfer(a, b, distance_weight=0.05, eps=1e-12, use_cuda=True):
    if a.size(0) == 0:
        return torch.zeros((0,))
    d1, d2, _, _ = O.chamfer_forward(_np(a), _np(b))
    d1, d2 = torch.from_numpy(d1).clamp(min=eps).sqrt(), torch.from_numpy(d2).clamp(min=eps).sqrt()
    return (d1.mean(-1) + d2.mean(-1)) / 2.0 * distance_weight


class SegmentGIoULossReference(torch.autograd.Function):
    """The plain tensor-op composition of the per-segment GIoU loss (iou_loss.py:69-129 with reduction 'mean' per segment):
    the CPU stand-in, and the reference the GPU test holds `train_ops._SegmentGIoULoss` against.  `giou_fn`: the
    convex_giou to use (class attribute, set by the caller)."""
    giou_fn = None

    @staticmethod
    def forward(ctx, pred, target, weight, seg, nseg, denom, loss_weight):
        if pred.size(0) == 0:
            ctx.save_for_backward(torch.zeros_like(pred))
            return pred.new_zeros((nseg,))
        gious, grad = SegmentGIoULossReference.giou_fn(pred, target)
        w = weight.to(gious.dtype)
        d = denom.to(gious.dtype).clamp(min=1.0)
        loss = torch.zeros((nseg,), dtype=gious.dtype, device=gious.device).index_add_(0, seg, (1 - gious) * w) / d
        unvalid = (grad > 1).sum(1) > 0
        grad = torch.where(unvalid[:, None], torch.full_like(grad, 1e-6), grad)
        ctx.save_for_backward(-grad * (w / d[seg])[:, None] * loss_weight)
        return loss * loss_weight

    @staticmethod
    def backward(ctx, grad_out=None):
        return ctx.saved_tensors[0], None, None, None, None, None, None


def segment_border_loss_reference(pts, gt, weight, seg, nseg, denom, loss_weight, inside_fn):
    """SpatialBorderLoss per segment as plain differentiable tensor operations (spatial_border_loss.py:8-92)."""
    P = pts.size(0)
    out = pts.new_zeros((nseg,))
    if P == 0:
        return out
    inside = inside_fn(pts.detach(), gt)
    outside = (inside == 0) & (weight > 0)[:, None]
    p9 = pts.reshape(P, 9, 2)
    centre = torch.stack([(gt[:, 0] + gt[:, 4]) / 2.0, (gt[:, 1] + gt[:, 5]) / 2.0], 1)[:, None, :]
    d2 = ((p9 - centre) ** 2).sum(-1)
    dist = 0.2 * torch.where(outside, d2, torch.ones_like(d2)).sqrt()
    dist = torch.where(outside, dist, torch.zeros_like(dist))
    s_sum = out.index_add(0, seg, dist.sum(1))
    n_out = out.index_add(0, seg, outside.sum(1).to(out.dtype))
    return loss_weight * (s_sum / n_out.clamp(min=1.0)) / (denom.to(out.dtype) + 1e-6)


class _CpuSegmentGIoU:
    @staticmethod
    def apply(pred, target, weight, seg, nseg, denom, loss_weight):
        SegmentGIoULossReference.giou_fn = staticmethod(_convex_giou)
        return SegmentGIoULossReference.apply(pred, target, weight, seg, nseg, denom, loss_weight)


def _cpu_border(pts, gt, weight, seg, nseg, denom, loss_weight):
    return segment_border_loss_reference(pts, gt, weight, seg, nseg, denom, loss_weight,
                                         lambda p, q: torch.from_numpy(O.points_in_quad_aligned(_np(p), _np(q))))


@contextlib.contextmanager
def installed():
    """Patch the operator entry points the detector's training step reaches; restores them on exit."""
    m = importlib.import_module
    dc = m('orientedreppoints_amd.mmdet_ops.deform_conv')
    apaa = m('orientedreppoints_amd.mmdet_ops.apaa')
    tro = m('orientedreppoints_amd.mmdet_ops.train_ops')
    losses = m('orientedreppoints_amd.mmdet_models.losses')
    ht = m('orientedreppoints_amd.mmdet_models.orientedreppoints_head_train')
    asg = m('orientedreppoints_amd.mmdet_models.assigners')
    patches = [
        (dc, 'deform_conv', _CpuDeformConv.apply), (dc, 'modulated_deform_conv', _CpuModulatedDeformConv.apply),
        (apaa, 'point_assign', _point_assign), (apaa, 'max_iou_assign', _max_iou_assign),
        (apaa, 'apaa_feature_dissimilarity', _feature_dissimilarity), (apaa, 'apaa_select', _apaa_select),
        (tro, 'pointset_target', _pointset_target), (tro, 'points_from_offsets', _points_from_offsets),
        (tro, 'gather_levels', _gather_levels), (tro, 'outline_samples', _outline_samples),
        (losses, '_sigmoid_focal_loss', _focal), (losses, 'convex_giou', _convex_giou),
        (losses, 'points_in_quad_aligned', lambda p, q: torch.from_numpy(O.points_in_quad_aligned(_np(p), _np(q)))),
        (ht, 'convex_giou', _convex_giou), (ht, 'ChamferDistance2D', _chamfer),
        (ht, '_SegmentGIoULoss', _CpuSegmentGIoU), (ht, '_segment_border_loss', _cpu_border),
        (ht, 'minaerarect', lambda p: torch.from_numpy(O.minarearect(_np(p)).astype(np.float32)) if p.numel() else torch.zeros((0, 8))),
        (asg, 'convex_iou', _convex_iou),
    ]
    saved = [(mod, name, getattr(mod, name)) for mod, name, _ in patches]
    for mod, name, fn in patches:
        setattr(mod, name, fn)
    try:
        yield
    finally:
        for mod, name, fn in saved:
            setattr(mod, name, fn)
