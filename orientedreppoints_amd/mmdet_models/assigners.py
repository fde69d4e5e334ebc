"""PointAssigner / MaxIoUAssigner / AssignResult / PseudoSampler / SamplingResult -- mirrors of
mmdet/core/bbox/assigners/{point_assigner.py:7-145, max_iou_assigner.py:7-152, assign_result.py} and
mmdet/core/bbox/samplers/{pseudo_sampler.py:6-25, sampling_result.py:24-48} with the same constructor arguments and
result fields; the per-gt Python loops are replaced by the device kernels of csrc/orp_assign.hip."""
import torch

from ..mmdet_ops import apaa
from ..mmdet_ops.iou_wrapper import convex_iou
from .registry import BBOX_ASSIGNERS


class AssignResult(object):

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels

    @property
    def num_preds(self):
        return len(self.gt_inds)


def _labels_from_gt_inds(assigned_gt_inds, gt_labels):
    if gt_labels is None:
        return None
    pos = assigned_gt_inds > 0
    idx = (assigned_gt_inds - 1).clamp(min=0)
    return torch.where(pos, gt_labels[idx], torch.zeros_like(assigned_gt_inds))


@BBOX_ASSIGNERS.register_module
class PointAssigner(object):
    """Each gt takes the `pos_num` nearest points of its pyramid level; 0 = negative, i+1 = gt i."""

    def __init__(self, scale=4, pos_num=3):
        self.scale = scale
        self.pos_num = pos_num

    def assign(self, points, gt_rbboxes, gt_rbboxes_ignore=None, gt_labels=None):
        num_points, num_gts = points.shape[0], gt_rbboxes.shape[0]
        if num_gts == 0 or num_points == 0:
            assigned_gt_inds = points.new_full((num_points,), 0, dtype=torch.long)
            assigned_labels = None if gt_labels is None else points.new_zeros((num_points,), dtype=torch.long)
            return AssignResult(num_gts, assigned_gt_inds, None, labels=assigned_labels)
        assert gt_rbboxes.size(1) == 8, 'gt_rbboxes should be (N * 8)'
        assigned_gt_inds = apaa.point_assign(points, gt_rbboxes, self.scale, self.pos_num)
        return AssignResult(num_gts, assigned_gt_inds, None, labels=_labels_from_gt_inds(assigned_gt_inds, gt_labels))


@BBOX_ASSIGNERS.register_module
class MaxIoUAssigner(object):
    """-1 don't care / 0 negative / i+1 gt i, from IoU(hull(9 points), gt) (max_iou_assigner.py:66 convex_overlaps)."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, gpu_assign_thr=-1):
        self.pos_iou_thr = pos_iou_thr
        self.neg_iou_thr = neg_iou_thr
        self.min_pos_iou = min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr = ignore_iof_thr
        self.ignore_wrt_candidates = ignore_wrt_candidates
        self.gpu_assign_thr = gpu_assign_thr

    def assign(self, points, gt_rbboxes, gt_rbboxes_ignore=None, gt_labels=None):
        if self.ignore_iof_thr > 0 and gt_rbboxes_ignore is not None and gt_rbboxes_ignore.numel() > 0:
            raise NotImplementedError('ignore regions are not used by the DOTA configs (ignore_iof_thr=-1)')
        num_gts, num_pts = gt_rbboxes.shape[0], points.shape[0]
        if num_gts == 0 or num_pts == 0:
            return self.assign_wrt_overlaps(points.new_zeros((num_gts, num_pts)), gt_labels)
        overlaps_nk = convex_iou(points, gt_rbboxes)          # [N, K] stays point-major on the device
        return self._assign_nk(overlaps_nk, gt_labels)

    def _assign_nk(self, overlaps_nk, gt_labels):
        num_gts = overlaps_nk.size(1)
        gt_inds, max_overlaps = apaa.max_iou_assign(overlaps_nk, self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou,
                                                    self.gt_max_assign_all)
        return AssignResult(num_gts, gt_inds, max_overlaps, labels=_labels_from_gt_inds(gt_inds, gt_labels))

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        """overlaps [K, N] (the reference's orientation)."""
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        if num_gts == 0 or num_bboxes == 0:
            assigned_gt_inds = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            if num_gts == 0:
                assigned_gt_inds[:] = 0
            assigned_labels = None if gt_labels is None else overlaps.new_zeros((num_bboxes,), dtype=torch.long)
            return AssignResult(num_gts, assigned_gt_inds, max_overlaps, labels=assigned_labels)
        return self._assign_nk(overlaps.t().contiguous(), gt_labels)


class SamplingResult(object):

    def __init__(self, pos_inds, neg_inds, points, gt_rbboxes, assign_result, gt_flags):
        self.pos_inds = pos_inds
        self.neg_inds = neg_inds
        self.pos_points = points[pos_inds]
        self.neg_points = points[neg_inds]
        self.pos_is_gt = gt_flags[pos_inds]
        self.num_gts = gt_rbboxes.shape[0]
        self.pos_assigned_gt_inds = assign_result.gt_inds[pos_inds] - 1
        if gt_rbboxes.numel() == 0:
            assert self.pos_assigned_gt_inds.numel() == 0
            self.pos_gt_rbboxes = torch.empty_like(gt_rbboxes).view(-1, 8)
        else:
            if len(gt_rbboxes.shape) < 2:
                gt_rbboxes = gt_rbboxes.view(-1, 8)
            self.pos_gt_rbboxes = gt_rbboxes[self.pos_assigned_gt_inds, :]
        self.pos_gt_labels = assign_result.labels[pos_inds] if assign_result.labels is not None else None


class PseudoSampler(object):

    def __init__(self, **kwargs):
        pass

    def sample(self, assign_result, bboxes, gt_bboxes, **kwargs):
        pos_inds = torch.nonzero(assign_result.gt_inds > 0).squeeze(-1)
        neg_inds = torch.nonzero(assign_result.gt_inds == 0).squeeze(-1)
        gt_flags = bboxes.new_zeros(bboxes.shape[0], dtype=torch.uint8)
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)
