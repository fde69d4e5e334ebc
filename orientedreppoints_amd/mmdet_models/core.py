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


def rbbox2result(bboxes, labels, num_classes):
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 9), dtype=np.float32) for _ in range(num_classes - 1)]
    bboxes = bboxes.cpu().numpy()
    labels = labels.cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes - 1)]
