"""Losses of the dense head -- mirrors of mmdet/models/losses/{focal_loss.py:71-108, iou_loss.py:69-129,
spatial_border_loss.py:8-92, utils.py:6-52} on the HIP operators.

Re-designed: SpatialBorderLoss asks the aligned point-in-quad kernel for the [P,9] flags directly instead of building
nine [P,P] matrices to read their diagonals (spatial_border_loss.py:24-67)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ..mmdet_ops.iou_wrapper import convex_giou
from ..mmdet_ops.point_justify import points_in_quad_aligned
from ..mmdet_ops.sigmoid_focal_loss import sigmoid_focal_loss as _sigmoid_focal_loss
from .registry import LOSSES


def reduce_loss(loss, reduction):
    reduction_enum = F._Reduction.get_enum(reduction)
    if reduction_enum == 0:
        return loss
    elif reduction_enum == 1:
        return loss.mean()
    return loss.sum()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        loss = reduce_loss(loss, reduction)
    else:
        if reduction == 'mean':
            loss = loss.sum() / avg_factor
        elif reduction != 'none':
            raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction='mean', avg_factor=None):
    loss = _sigmoid_focal_loss(pred, target, gamma, alpha)
    if weight is not None:
        weight = weight.view(-1, 1)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module
class FocalLoss(nn.Module):

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction='mean', loss_weight=1.0):
        super(FocalLoss, self).__init__()
        assert use_sigmoid is True, 'Only sigmoid focal loss supported now.'
        self.use_sigmoid = use_sigmoid
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * sigmoid_focal_loss(pred, target, weight, gamma=self.gamma, alpha=self.alpha,
                                                     reduction=reduction, avg_factor=avg_factor)


class GIoULossFuction(Function):
    """Loss AND gradient come out of one kernel call (iou_loss.py:69-100).  The reference's backward returns the stashed
    gradient (loss_weight folded in) and ignores the incoming one: that is what `convex_giou_loss(...)` does here too
    (`chain=False`).  The GIoULoss module calls it with `chain=True`: the stash then carries no loss_weight and is multiplied
    by the incoming gradient, which is loss_weight (the module's outer product) times whatever is upstream -- the same bits
    as the reference whenever upstream is 1 (every use in the reference: the losses are summed and `.backward()` is called
    on the sum), and correct under fp16 loss scaling (`GradScaler.scale(loss).backward()`), where ignoring it would leave
    the two localisation losses unscaled and then divide them by the scale."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)   # under autocast: fp32 inputs, autocast off inside
    def forward(ctx, pred, target, weight=None, reduction=None, avg_factor=None, loss_weight=1.0, chain=False):
        ctx.save_for_backward(pred)
        convex_gious, grad = convex_giou(pred, target)
        loss = 1 - convex_gious
        if weight is not None:
            loss = loss * weight
            grad = grad * weight.reshape(-1, 1)
        if reduction == 'sum':
            loss = loss.sum()
        elif reduction == 'mean':
            loss = loss.mean()
        # rows with any gradient component > 1 are replaced by 1e-6 (iou_loss.py:87-89)
        unvalid = (grad > 1).sum(1) > 0
        grad = torch.where(unvalid[:, None], torch.full_like(grad, 1e-6), grad)
        ctx.chain = bool(chain)
        ctx.convex_points_grad = -grad / grad.size(0) * (1.0 if chain else loss_weight)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out=None):
        g = ctx.convex_points_grad
        if ctx.chain and grad_out is not None:
            go = grad_out.to(g.dtype)
            g = g * (go.reshape(()) if go.numel() == 1 else go.reshape(-1, 1))   # scalar (mean / sum) or per row (none)
        return g, None, None, None, None, None, None


convex_giou_loss = GIoULossFuction.apply


@LOSSES.register_module
class GIoULoss(nn.Module):

    def __init__(self, reduction='mean', loss_weight=1.0):
        super(GIoULoss, self).__init__()
        self.reduction = reduction
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            return (pred * weight.unsqueeze(-1)).sum()  # 0
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        return self.loss_weight * convex_giou_loss(pred, target, weight, reduction, avg_factor, self.loss_weight, True)


def spatial_border_loss(pts, gt_bboxes, reduction='mean', y_first=False):
    num_gt, num_pts = gt_bboxes.size(0), pts.size(0)
    loss = pts.new_zeros([0])
    if num_gt > 0:
        inside_flag = points_in_quad_aligned(pts, gt_bboxes)          # [P, 9]: 1 inside, 0 outside / on the border
        pts = pts.reshape(-1, 9, 2)
        outside = torch.where(inside_flag == 0)
        out_border_pts = pts[outside]
        if out_border_pts.size(0) > 0:
            corres_gt_boxes = gt_bboxes[outside[0]]
            cx = (corres_gt_boxes[:, 0] + corres_gt_boxes[:, 4]) / 2.0
            cy = (corres_gt_boxes[:, 1] + corres_gt_boxes[:, 5]) / 2.0
            center = torch.stack([cx, cy], dim=1)
            distance_out_pts = 0.2 * (((out_border_pts - center) ** 2).sum(dim=1).sqrt())
            loss = distance_out_pts.sum() / out_border_pts.size(0)
    return loss


def weighted_spatial_border_loss(pts, gt_bboxes, weight, avg_factor=None, y_first=False):
    weight = weight.unsqueeze(dim=1).repeat(1, 4)
    assert weight.dim() == 2
    if avg_factor is None:
        avg_factor = torch.sum(weight > 0).float().item() / 4 + 1e-6
    loss = spatial_border_loss(pts, gt_bboxes, y_first=y_first, reduction='none')
    return torch.sum(loss)[None] / avg_factor


@LOSSES.register_module
class SpatialBorderLoss(nn.Module):

    def __init__(self, loss_weight=1.0):
        super(SpatialBorderLoss, self).__init__()
        self.loss_weight = loss_weight

    def forward(self, pts, gt_bboxes, weight, y_first=False, *args, **kwargs):
        return self.loss_weight * weighted_spatial_border_loss(pts, gt_bboxes, weight, y_first=y_first, *args,
                                                               **kwargs)
