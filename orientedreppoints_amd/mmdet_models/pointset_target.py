"""init_pointset_target / refine_pointset_target (mmdet/core/bbox/pointset_target.py:6-230): per-image target
building for both stages.  Same outputs as the reference (labels, label weights, rbbox_gt [N,8], proposal weights,
pos/neg indices, gt indices, unmapped to the full point set), with the assigners running on the device kernels."""
import torch

from .assigners import PseudoSampler
from .core import multi_apply, unmap
from .registry import build_assigner


def images_to_levels(target, num_level_anchors):
    target = torch.stack(target, 0)
    level_targets = []
    start = 0
    for n in num_level_anchors:
        end = start + n
        level_targets.append(target[:, start:end].squeeze(0))
        start = end
    return level_targets


def _target_single(flat_proposals, valid_flags, gt_rbboxes, gt_rbboxes_ignore, gt_labels, cfg, label_channels=1,
                   sampling=True, unmap_outputs=True):
    inside_flags = valid_flags
    if not inside_flags.any():
        return (None,) * 8
    assert not sampling, 'the focal-loss configs use PseudoSampler (sampling=False)'
    proposals = flat_proposals[inside_flags, :]
    bbox_assigner = build_assigner(cfg.assigner)
    assign_result = bbox_assigner.assign(proposals, gt_rbboxes, gt_rbboxes_ignore, gt_labels)
    sampling_result = PseudoSampler().sample(assign_result, proposals, gt_rbboxes)
    gt_inds = assign_result.gt_inds
    num_valid_proposals = proposals.shape[0]
    rbbox_gt = proposals.new_zeros([num_valid_proposals, 8])
    pos_proposals = torch.zeros_like(proposals)
    proposals_weights = proposals.new_zeros(num_valid_proposals)
    labels = proposals.new_zeros(num_valid_proposals, dtype=torch.long)
    label_weights = proposals.new_zeros(num_valid_proposals, dtype=torch.float)
    pos_inds = sampling_result.pos_inds
    neg_inds = sampling_result.neg_inds
    if len(pos_inds) > 0:
        rbbox_gt[pos_inds, :] = sampling_result.pos_gt_rbboxes
        pos_proposals[pos_inds, :] = proposals[pos_inds, :]
        proposals_weights[pos_inds] = 1.0
        if gt_labels is None:
            labels[pos_inds] = 1
        else:
            labels[pos_inds] = gt_labels[sampling_result.pos_assigned_gt_inds]
        label_weights[pos_inds] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    if unmap_outputs:
        num_total_proposals = flat_proposals.size(0)
        labels = unmap(labels, num_total_proposals, inside_flags)
        label_weights = unmap(label_weights, num_total_proposals, inside_flags)
        rbbox_gt = unmap(rbbox_gt, num_total_proposals, inside_flags)
        pos_proposals = unmap(pos_proposals, num_total_proposals, inside_flags)
        proposals_weights = unmap(proposals_weights, num_total_proposals, inside_flags)
        gt_inds = unmap(gt_inds, num_total_proposals, inside_flags)
    return (labels, label_weights, rbbox_gt, pos_proposals, proposals_weights, pos_inds, neg_inds, gt_inds)


init_pointset_target_single = _target_single
refine_pointset_target_single = _target_single


def _flatten(proposals_list, valid_flag_list, num_imgs):
    for i in range(num_imgs):
        assert len(proposals_list[i]) == len(valid_flag_list[i])
        proposals_list[i] = torch.cat(proposals_list[i])
        valid_flag_list[i] = torch.cat(valid_flag_list[i])


def init_pointset_target(proposals_list, valid_flag_list, gt_rbboxes_list, img_metas, cfg,
                         gt_rbboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                         unmap_outputs=True):
    num_imgs = len(img_metas)
    assert len(proposals_list) == len(valid_flag_list) == num_imgs
    num_level_proposals = [points.size(0) for points in proposals_list[0]]
    _flatten(proposals_list, valid_flag_list, num_imgs)
    if gt_rbboxes_ignore_list is None:
        gt_rbboxes_ignore_list = [None for _ in range(num_imgs)]
    if gt_labels_list is None:
        gt_labels_list = [None for _ in range(num_imgs)]
    (all_labels, all_label_weights, all_rbbox_gt, all_proposals, all_proposal_weights, pos_inds_list, neg_inds_list,
     all_gt_inds_list) = multi_apply(_target_single, proposals_list, valid_flag_list, gt_rbboxes_list,
                                     gt_rbboxes_ignore_list, gt_labels_list, cfg=cfg, label_channels=label_channels,
                                     sampling=sampling, unmap_outputs=unmap_outputs)
    if any([labels is None for labels in all_labels]):
        return None
    num_total_pos = sum([max(inds.numel(), 1) for inds in pos_inds_list])
    num_total_neg = sum([max(inds.numel(), 1) for inds in neg_inds_list])
    return (images_to_levels(all_labels, num_level_proposals), images_to_levels(all_label_weights, num_level_proposals),
            images_to_levels(all_rbbox_gt, num_level_proposals), images_to_levels(all_proposals, num_level_proposals),
            images_to_levels(all_proposal_weights, num_level_proposals), num_total_pos, num_total_neg,
            images_to_levels(all_gt_inds_list, num_level_proposals))


def refine_pointset_target(proposals_list, valid_flag_list, gt_rbboxes_list, img_metas, cfg,
                           gt_rbboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                           unmap_outputs=True):
    num_imgs = len(img_metas)
    assert len(proposals_list) == len(valid_flag_list) == num_imgs
    _flatten(proposals_list, valid_flag_list, num_imgs)
    if gt_rbboxes_ignore_list is None:
        gt_rbboxes_ignore_list = [None for _ in range(num_imgs)]
    if gt_labels_list is None:
        gt_labels_list = [None for _ in range(num_imgs)]
    (all_labels, all_label_weights, all_rbbox_gt, all_proposals, all_proposal_weights, pos_inds_list, neg_inds_list,
     all_gt_inds) = multi_apply(_target_single, proposals_list, valid_flag_list, gt_rbboxes_list,
                                gt_rbboxes_ignore_list, gt_labels_list, cfg=cfg, label_channels=label_channels,
                                sampling=sampling, unmap_outputs=unmap_outputs)
    pos_inds, pos_gt_index = [], []
    for i, single_labels in enumerate(all_labels):
        idx = (single_labels > 0).nonzero().view(-1)
        pos_inds.append(idx)
        pos_gt_index.append(all_gt_inds[i][idx])
    return (all_labels, all_label_weights, all_rbbox_gt, all_proposals, all_proposal_weights, pos_inds, pos_gt_index)
