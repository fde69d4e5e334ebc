"""Training side of OrientedRepPointsHead: loss() with the APAA adaptive points assessment and assignment
(mmdet/models/anchor_heads/orientedreppoints_head.py:176-222 get_points / offset_to_pts, :250-292 sampling_points,
:294-318 init_loss_single, :320-493 loss, :495-520 get_adaptive_points_feature, :522-573 points_quality_assessment,
:576-600 feature_cosine_similarity, :602-671 point_samples_selection).

Same inputs, same five loss terms.  What is re-designed for the MI355X build:
  * the adaptive point features are sampled for the POSITIVES only and reduced to the dissimilarity score in one
    kernel (the reference grid_samples all N x 9 points into a [B,256,N,9] fp32 buffer, 201 MB per image, and reads
    back the positives);
  * point_samples_selection's Python loop over gts x levels (mask + topk + cat per iteration) is one kernel launch;
  * both assigners run as device kernels (assigners.py);  SpatialBorderLoss asks for the aligned [P,9] flags.
"""
import numpy as np
import torch

from ..mmdet_ops import apaa
from ..mmdet_ops.chamfer_distance import ChamferDistance2D
from ..mmdet_ops.minarea_rect import minaerarect
from .core import levels_to_images, multi_apply
from .pointset_target import init_pointset_target, refine_pointset_target


def get_points(head, featmap_sizes, img_metas, device):
    num_imgs = len(img_metas)
    num_levels = len(featmap_sizes)
    multi_level_points = [head.point_generators[i].grid_points(featmap_sizes[i], head.point_strides[i], device)
                          for i in range(num_levels)]
    points_list = [[point.clone() for point in multi_level_points] for _ in range(num_imgs)]
    valid_flag_list = []
    for img_meta in img_metas:
        multi_level_flags = []
        for i in range(num_levels):
            point_stride = head.point_strides[i]
            feat_h, feat_w = featmap_sizes[i]
            h, w = img_meta['pad_shape'][:2]
            valid_feat_h = min(int(np.ceil(h / point_stride)), feat_h)
            valid_feat_w = min(int(np.ceil(w / point_stride)), feat_w)
            multi_level_flags.append(head.point_generators[i].valid_flags((feat_h, feat_w),
                                                                          (valid_feat_h, valid_feat_w), device))
        valid_flag_list.append(multi_level_flags)
    return points_list, valid_flag_list


def offset_to_pts(head, center_list, pred_list):
    """[lvl][B,18,H,W] (y,x) grid-unit offsets -> [lvl][B,HW,18] (x,y) image coordinates (head :204-222)."""
    pts_list = []
    for i_lvl in range(len(head.point_strides)):
        pred = pred_list[i_lvl]
        B = pred.size(0)
        yx = pred.permute(0, 2, 3, 1).reshape(B, -1, head.num_points, 2)
        xy = yx.flip(-1).reshape(B, -1, 2 * head.num_points)
        centers = torch.stack([center_list[i][i_lvl][:, :2].repeat(1, head.num_points) for i in range(B)], 0)
        pts_list.append(xy * head.point_strides[i_lvl] + centers)
    return pts_list


def sampling_points(corners, points_num):
    """10 linspace(0,1) points on each edge 1->2->3->4->1 of [P,8] corners -> [P, 4*points_num, 2] (head :250-292)."""
    c = corners.reshape(-1, 4, 2)
    nxt = torch.roll(c, shifts=-1, dims=1)
    ratio = torch.linspace(0, 1, points_num, device=corners.device).view(1, 1, points_num, 1)
    pts = ratio * nxt.unsqueeze(2) + (1 - ratio) * c.unsqueeze(2)          # [P,4,n,2]
    return pts.reshape(c.size(0), 4 * points_num, 2)


def init_loss_single(head, pts_pred_init, rbox_gt_init, rbox_weights_init, stride):
    normalize_term = head.point_base_scale * stride
    rbox_gt_init = rbox_gt_init.reshape(-1, 8)
    rbox_weights_init = rbox_weights_init.reshape(-1)
    pts_pred_init = pts_pred_init.reshape(-1, 2 * head.num_points)
    pos_ind_init = (rbox_weights_init > 0).nonzero().reshape(-1)
    pts_pred_init_norm = pts_pred_init[pos_ind_init]
    rbox_gt_init_norm = rbox_gt_init[pos_ind_init]
    rbox_weights_pos_init = rbox_weights_init[pos_ind_init]
    loss_rbox_init = head.loss_rbox_init(pts_pred_init_norm / normalize_term, rbox_gt_init_norm / normalize_term,
                                         rbox_weights_pos_init)
    loss_border_init = head.loss_spatial_init(
        pts_pred_init_norm.reshape(-1, 2 * head.num_points) / normalize_term, rbox_gt_init_norm / normalize_term,
        rbox_weights_pos_init, y_first=False, avg_factor=None
    ) if head.loss_spatial_init is not None else loss_rbox_init.new_zeros(1)
    return loss_rbox_init, loss_border_init


def points_quality_assessment(head, feats, img_id, level_of_index, cls_score, pts_pred_init, pts_pred_refine, label,
                              rbbox_gt, label_weight, rbox_weight, pos_inds):
    """APAA quality Q of every positive of one image (head :522-573)."""
    pos_scores = cls_score[pos_inds]
    pos_pts_pred_init = pts_pred_init[pos_inds]
    pos_pts_pred_refine = pts_pred_refine[pos_inds]
    pos_rbbox_gt = rbbox_gt[pos_inds]
    pos_label = label[pos_inds]
    pos_label_weight = label_weight[pos_inds]
    pos_rbox_weight = rbox_weight[pos_inds]
    P = pos_inds.numel()
    img_index = torch.full((P,), img_id, dtype=torch.int32, device=pos_inds.device)
    pts_feats_dissimilarity = apaa.apaa_feature_dissimilarity(feats, head.point_strides, pos_pts_pred_refine,
                                                              img_index, level_of_index[pos_inds])
    qua_cls = head.loss_cls(pos_scores, pos_label, pos_label_weight, avg_factor=head.loss_cls.loss_weight,
                            reduction_override='none')
    corners_pred_init = minaerarect(pos_pts_pred_init)
    corners_pred_refine = minaerarect(pos_pts_pred_refine)
    sampling_pts_pred_init = sampling_points(corners_pred_init, 10)
    sampling_pts_pred_refine = sampling_points(corners_pred_refine, 10)
    corners_pts_gt = sampling_points(pos_rbbox_gt, 10)
    qua_ori_init = ChamferDistance2D(corners_pts_gt, sampling_pts_pred_init)
    qua_ori_refine = ChamferDistance2D(corners_pts_gt, sampling_pts_pred_refine)
    qua_loc_init = head.loss_rbox_refine(pos_pts_pred_init, pos_rbbox_gt, pos_rbox_weight,
                                         avg_factor=head.loss_cls.loss_weight, reduction_override='none')
    qua_loc_refine = head.loss_rbox_refine(pos_pts_pred_refine, pos_rbbox_gt, pos_rbox_weight,
                                           avg_factor=head.loss_cls.loss_weight, reduction_override='none')
    qua_cls = qua_cls.sum(-1)
    qua = qua_cls + 0.2 * (qua_loc_init + 0.3 * qua_ori_init) + 0.8 * (qua_loc_refine + 0.3 * qua_ori_refine) \
        + 0.1 * pts_feats_dissimilarity
    return qua


def point_samples_selection(head, quality_assess, label, label_weight, rbox_weight, pos_inds, pos_gt_inds,
                            level_of_index, num_level, num_gt):
    """Keep, per gt, the top_ratio best of the per-level 6 best positives; the rest become background (head :602-671).
    Returns (label, label_weight, rbox_weight, num_pos [0-d tensor], pos_normalize_term in ascending-index order)."""
    if pos_inds.numel() == 0:
        return label, label_weight, rbox_weight, label.new_zeros(()), rbox_weight.new_zeros((0,))
    pos_level = level_of_index[pos_inds]
    keep = apaa.apaa_select(quality_assess, pos_gt_inds, pos_level, num_gt, num_level, 6, head.top_ratio)
    reassign_ids = pos_inds[~keep]
    label[reassign_ids] = 0
    rbox_weight[reassign_ids] = 0
    kept_inds = pos_inds[keep]                                     # ascending, as (labels > 0).nonzero() will be
    strides = torch.as_tensor(head.point_strides, dtype=rbox_weight.dtype, device=rbox_weight.device)
    pos_normalize_term = head.point_base_scale * strides[level_of_index[kept_inds].long()]
    return label, label_weight, rbox_weight, keep.sum(), pos_normalize_term


def head_loss(head, cls_scores, pts_preds_init, pts_preds_refine, base_features, gt_rbboxes, gt_labels, img_metas, cfg,
              gt_rbboxes_ignore=None, record=None):
    """`record` (tests only): a dict that receives the intermediate targets (init / refine assignment, APAA quality and
    selection per image) so that they can be compared one by one with the reference's."""
    featmap_sizes = [featmap.size()[-2:] for featmap in cls_scores]
    assert len(featmap_sizes) == len(head.point_generators)
    device = cls_scores[0].device
    label_channels = head.cls_out_channels if head.use_sigmoid_cls else 1
    num_level = len(featmap_sizes)
    num_imgs = len(img_metas)

    # ---- init stage targets ---------------------------------------------------------------------------------------
    center_list, valid_flag_list = get_points(head, featmap_sizes, img_metas, device)
    pts_coordinate_preds_init = offset_to_pts(head, center_list, pts_preds_init)
    num_proposals_each_level = [int(fs[0] * fs[1]) for fs in featmap_sizes]
    cls_reg_targets_init = init_pointset_target(center_list, valid_flag_list, gt_rbboxes, img_metas, cfg.init,
                                                gt_rbboxes_ignore_list=gt_rbboxes_ignore, gt_labels_list=gt_labels,
                                                label_channels=label_channels, sampling=head.sampling)
    (*_, rbbox_gt_list_init, candidate_list_init, rbox_weights_list_init, num_total_pos_init, num_total_neg_init,
     gt_inds_init) = cls_reg_targets_init

    if record is not None:
        record['init_target'] = cls_reg_targets_init

    # ---- refine stage targets: the init-stage point sets (detached) are the proposals ----------------------------
    center_list, valid_flag_list = get_points(head, featmap_sizes, img_metas, device)
    pts_coordinate_preds_refine = offset_to_pts(head, center_list, pts_preds_refine)
    # NB reference quirk kept for parity (head :378-381): the refine-stage proposals add the init offsets to the
    # (x, y) centres WITHOUT the (y, x) -> (x, y) swap that offset_to_pts applies.
    points_list = []
    for i_img, center in enumerate(center_list):
        points = []
        for i_lvl in range(num_level):
            pred = pts_preds_init[i_lvl].detach()
            shift = pred.permute(0, 2, 3, 1) * head.point_strides[i_lvl]
            points_center = center[i_lvl][:, :2].repeat(1, head.num_points)
            points.append(points_center + shift[i_img].reshape(-1, 2 * head.num_points))
        points_list.append(points)
    cls_reg_targets_refine = refine_pointset_target(points_list, valid_flag_list, gt_rbboxes, img_metas, cfg.refine,
                                                    gt_rbboxes_ignore_list=gt_rbboxes_ignore,
                                                    gt_labels_list=gt_labels, label_channels=label_channels,
                                                    sampling=head.sampling)
    (labels_list, label_weights_list, rbox_gt_list_refine, _, rbox_weights_list_refine, pos_inds_list_refine,
     pos_gt_index_list_refine) = cls_reg_targets_refine

    if record is not None:
        record['refine_target'] = [[t.clone() if torch.is_tensor(t) else t for t in lst] for lst in cls_reg_targets_refine]
        record['qa'], record['sel'] = [], []

    cls_scores = levels_to_images(cls_scores)
    cls_scores = [item.reshape(-1, head.cls_out_channels) for item in cls_scores]
    pts_init_img = [item.reshape(-1, 2 * head.num_points)
                    for item in levels_to_images(pts_coordinate_preds_init, flatten=True)]
    pts_refine_img = [item.reshape(-1, 2 * head.num_points)
                      for item in levels_to_images(pts_coordinate_preds_refine, flatten=True)]
    level_of_index = torch.cat([torch.full((n,), l, dtype=torch.int32, device=device)
                                for l, n in enumerate(num_proposals_each_level)])

    # ---- APAA: quality assessment + sample selection (no gradient) -------------------------------------------------
    with torch.no_grad():
        feats = [f.detach() for f in base_features]
        num_pos = 0
        pos_normalize_terms = []
        for i in range(num_imgs):
            pos_inds = pos_inds_list_refine[i]
            if pos_inds.numel() > 0:
                qua = points_quality_assessment(head, feats, i, level_of_index, cls_scores[i], pts_init_img[i],
                                                pts_refine_img[i], labels_list[i], rbox_gt_list_refine[i],
                                                label_weights_list[i], rbox_weights_list_refine[i], pos_inds)
            else:
                qua = rbox_weights_list_refine[i].new_zeros((0,))
            (labels_list[i], label_weights_list[i], rbox_weights_list_refine[i], npos_i, pnt) = point_samples_selection(
                head, qua, labels_list[i], label_weights_list[i], rbox_weights_list_refine[i], pos_inds,
                pos_gt_index_list_refine[i], level_of_index, num_level, int(gt_rbboxes[i].shape[0]))
            num_pos = num_pos + npos_i
            pos_normalize_terms.append(pnt)
            if record is not None:
                record['qa'].append(qua.clone())
                record['sel'].append((labels_list[i].clone(), label_weights_list[i].clone(),
                                      rbox_weights_list_refine[i].clone(), int(npos_i), pnt.clone()))

    cls_scores = torch.cat(cls_scores, 0).view(-1, cls_scores[0].size(-1))
    pts_preds_refine_all = torch.cat(pts_refine_img, 0).view(-1, pts_refine_img[0].size(-1))
    labels = torch.cat(labels_list, 0).view(-1)
    labels_weight = torch.cat(label_weights_list, 0).view(-1)
    rbox_gt_refine = torch.cat(rbox_gt_list_refine, 0).view(-1, rbox_gt_list_refine[0].size(-1))
    rbox_weights_refine = torch.cat(rbox_weights_list_refine, 0).view(-1)
    pos_normalize_term = torch.cat(pos_normalize_terms, 0).reshape(-1)
    pos_inds_flatten = (labels > 0).nonzero().reshape(-1)
    assert len(pos_normalize_term) == len(pos_inds_flatten)
    num_pos = int(num_pos)                   # the reference's python `num_pos`; one host sync per step
    if num_pos:
        losses_cls = head.loss_cls(cls_scores, labels, labels_weight, avg_factor=num_pos)
        pos_pts_pred_refine = pts_preds_refine_all[pos_inds_flatten]
        pos_rbox_gt_refine = rbox_gt_refine[pos_inds_flatten]
        pos_rbox_weights_refine = rbox_weights_refine[pos_inds_flatten]
        losses_rbox_refine = head.loss_rbox_refine(pos_pts_pred_refine / pos_normalize_term.reshape(-1, 1),
                                                   pos_rbox_gt_refine / pos_normalize_term.reshape(-1, 1),
                                                   pos_rbox_weights_refine)
        loss_border_refine = head.loss_spatial_refine(
            pos_pts_pred_refine.reshape(-1, 2 * head.num_points) / pos_normalize_term.reshape(-1, 1),
            pos_rbox_gt_refine / pos_normalize_term.reshape(-1, 1), pos_rbox_weights_refine, y_first=False,
            avg_factor=None) if head.loss_spatial_refine is not None else losses_rbox_refine.new_zeros(1)
    else:
        losses_cls = cls_scores.sum() * 0
        losses_rbox_refine = pts_preds_refine_all.sum() * 0
        loss_border_refine = pts_preds_refine_all.sum() * 0

    losses_rbox_init, loss_border_init = multi_apply(
        lambda p, g, w, s: init_loss_single(head, p, g, w, s),
        pts_coordinate_preds_init, rbbox_gt_list_init, rbox_weights_list_init, head.point_strides)
    return {
        'loss_cls': losses_cls,
        'loss_rbox_init': losses_rbox_init,
        'loss_rbox_refine': losses_rbox_refine,
        'loss_spatial_init': loss_border_init,
        'loss_spatial_refine': loss_border_refine,
    }
