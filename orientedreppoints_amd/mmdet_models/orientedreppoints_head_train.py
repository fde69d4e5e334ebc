"""Training side of OrientedRepPointsHead: loss() with the APAA adaptive points assessment and assignment
(mmdet/models/anchor_heads/orientedreppoints_head.py:176-222 get_points / offset_to_pts, :250-292 sampling_points,
:294-318 init_loss_single, :320-493 loss, :495-520 get_adaptive_points_feature, :522-573 points_quality_assessment,
:576-600 feature_cosine_similarity, :602-671 point_samples_selection).

Same inputs, same five loss terms, same gradients.  What is re-designed for the MI355X build (round 3: the loss used to be
1 300 of the step's 3 100 kernel launches -- per-image / per-level loops of tiny tensor operations and nine host
synchronisations):
  * `head_loss` is BATCHED over images and levels.  Both stages' targets come from one `orp_pointset_target` launch each
    (pointset_target.py); the positives of all images are found with ONE host read of the per-image counts (then
    `nonzero_static`, which does not synchronise); the quality assessment, the selection and every loss term run once
    for all images / levels, with per-level and per-image means formed by segmented sums;
  * the head's outputs are only ever read at the positives: `gather_levels` (mmdet_ops/train_ops.py) picks those rows
    straight out of the [B,C,H,W] level tensors (and returns their gradient the same way), so the [B,N,18] point-set
    tensors of offset_to_pts and their backward passes are never materialised for autograd;
  * the adaptive point features are sampled for the positives only and reduced to the dissimilarity score in one kernel
    (the reference grid_samples all N x 9 points into a [B,256,N,9] fp32 buffer, 201 MB per image);
  * point_samples_selection's Python loop over gts x levels is one kernel launch over all images (global gt numbering).
The per-image / per-level functions of the reference (`offset_to_pts`, `sampling_points`, `points_quality_assessment`,
`point_samples_selection`, `init_loss_single`, `get_points`) keep their signatures on top of the same operators.
"""
import numpy as np
import torch

from ..mmdet_ops import apaa, train_ops
from ..mmdet_ops.chamfer_distance import ChamferDistance2D
from ..mmdet_ops.iou_wrapper import convex_giou
from ..mmdet_ops.minarea_rect import minaerarect
from .pointset_target import gt_tables, images_to_levels, pointset_targets


# ---- geometry of the point grid (constant per input shape: built once, reused every step) --------------------------------
class _Geometry(object):
    pass


def _geometry(head, featmap_sizes, img_metas, device):
    """centres [N,3] (x, y, stride) of all levels, valid [B,N] (None when every location of every image is valid -- known
    from the shapes alone), level / stride of every location, the levels' first locations."""
    key = (tuple((int(h), int(w)) for h, w in featmap_sizes), tuple(tuple(m['pad_shape'][:2]) for m in img_metas), str(device))
    cache = head.__dict__.setdefault('_train_geometry', {})
    g = cache.get(key)
    if g is not None:
        return g
    g = _Geometry()
    pts, lvl, strd, first, all_valid = [], [], [], [0], True
    flags = [[] for _ in img_metas]
    for i, (fh, fw) in enumerate(key[0]):
        s = head.point_strides[i]
        pts.append(head.point_generators[i].grid_points((fh, fw), s, device))
        lvl.append(torch.full((fh * fw,), i, dtype=torch.int32, device=device))
        strd.append(torch.full((fh * fw,), float(s), dtype=torch.float32, device=device))
        first.append(first[-1] + fh * fw)
        for b, m in enumerate(img_metas):
            h, w = m['pad_shape'][:2]
            vh, vw = min(int(np.ceil(h / s)), fh), min(int(np.ceil(w / s)), fw)
            all_valid = all_valid and vh == fh and vw == fw
            flags[b].append(head.point_generators[i].valid_flags((fh, fw), (vh, vw), device))
    g.centers = torch.cat(pts, 0)
    g.level = torch.cat(lvl)
    g.stride = torch.cat(strd)
    g.first = first
    g.N = first[-1]
    g.num_level = [b - a for a, b in zip(first[:-1], first[1:])]
    g.valid = None if all_valid else torch.stack([torch.cat(f) for f in flags], 0)
    if len(cache) >= 8:
        cache.pop(next(iter(cache)))
    cache[key] = g
    return g


def get_points(head, featmap_sizes, img_metas, device):
    """(points_list, valid_flag_list) per image and level (head :176-202).  The grid is shared, read-only data."""
    multi_level_points = [head.point_generators[i].grid_points(featmap_sizes[i], head.point_strides[i], device)
                          for i in range(len(featmap_sizes))]
    points_list = [list(multi_level_points) for _ in img_metas]
    valid_flag_list = []
    for img_meta in img_metas:
        flags = []
        for i, (feat_h, feat_w) in enumerate(featmap_sizes):
            s = head.point_strides[i]
            h, w = img_meta['pad_shape'][:2]
            flags.append(head.point_generators[i].valid_flags(
                (feat_h, feat_w), (min(int(np.ceil(h / s)), feat_h), min(int(np.ceil(w / s)), feat_w)), device))
        valid_flag_list.append(flags)
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
    return train_ops.outline_samples(corners, points_num)


# ---- losses over segments (levels of the init stage / the kept positives of the refine stage) ----------------------------
# mmdet_ops/train_ops.py: rows kernel + one fixed-order segment reduction each (csrc/orp_pointwise.hip)
_SegmentGIoULoss = train_ops._SegmentGIoULoss


def _segment_border_loss(pts, gt, weight, seg, nseg, denom, loss_weight):
    return train_ops.segment_border_loss(pts, gt, weight, seg, nseg, denom, loss_weight)


def init_loss_single(head, pts_pred_init, rbox_gt_init, rbox_weights_init, stride):
    """One level of the init-stage losses (head :294-318)."""
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


# ---- APAA quality and selection --------------------------------------------------------------------------------------------
def _quality(head, feats, img_index, level, pos_scores, pos_pts_init, pos_pts_refine, pos_label, pos_rbbox_gt,
             pos_label_weight, pos_rbox_weight):
    """Q of a set of positives (any mix of images): focal + 0.2 (L_giou + 0.3 CD)(init) + 0.8 (...)(refine) + 0.1 F."""
    diss = apaa.apaa_feature_dissimilarity(feats, head.point_strides, pos_pts_refine, img_index, level)
    qua_cls = head.loss_cls(pos_scores, pos_label, pos_label_weight, avg_factor=head.loss_cls.loss_weight,
                            reduction_override='none').sum(-1)
    gt_outline = sampling_points(pos_rbbox_gt, 10)
    qua_ori_init = ChamferDistance2D(gt_outline, sampling_points(minaerarect(pos_pts_init), 10))
    qua_ori_refine = ChamferDistance2D(gt_outline, sampling_points(minaerarect(pos_pts_refine), 10))
    # GIoULoss with reduction 'none' (iou_loss.py:69-129) = loss_weight * (1 - GIoU) * weight; called on the operator
    # directly: the module's `torch.any(weight > 0)` guard is a host synchronisation
    lw = head.loss_rbox_refine.loss_weight
    qua_loc_init = lw * (1 - convex_giou(pos_pts_init, pos_rbbox_gt)[0]) * pos_rbox_weight
    qua_loc_refine = lw * (1 - convex_giou(pos_pts_refine, pos_rbbox_gt)[0]) * pos_rbox_weight
    return qua_cls + 0.2 * (qua_loc_init + 0.3 * qua_ori_init) + 0.8 * (qua_loc_refine + 0.3 * qua_ori_refine) + 0.1 * diss


def points_quality_assessment(head, feats, img_id, level_of_index, cls_score, pts_pred_init, pts_pred_refine, label,
                              rbbox_gt, label_weight, rbox_weight, pos_inds):
    """APAA quality Q of every positive of one image (head :522-573)."""
    P = pos_inds.numel()
    img_index = torch.full((P,), img_id, dtype=torch.int32, device=pos_inds.device)
    return _quality(head, feats, img_index, level_of_index[pos_inds], cls_score[pos_inds], pts_pred_init[pos_inds],
                    pts_pred_refine[pos_inds], label[pos_inds], rbbox_gt[pos_inds], label_weight[pos_inds],
                    rbox_weight[pos_inds])


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


# ---- loss() -------------------------------------------------------------------------------------------------------------------
def head_loss(head, cls_scores, pts_preds_init, pts_preds_refine, base_features, gt_rbboxes, gt_labels, img_metas, cfg,
              gt_rbboxes_ignore=None, record=None):
    """`record` (tests only): a dict that receives the intermediate targets (init / refine assignment, APAA quality and
    selection per image) so that they can be compared one by one with the reference's."""
    featmap_sizes = [featmap.size()[-2:] for featmap in cls_scores]
    assert len(featmap_sizes) == len(head.point_generators)
    assert not head.sampling, 'the focal-loss configs use PseudoSampler (sampling=False)'
    device = cls_scores[0].device
    B, C = cls_scores[0].size(0), head.cls_out_channels
    num_level = len(featmap_sizes)
    strides = head.point_strides
    geo = _geometry(head, featmap_sizes, img_metas, device)
    N = geo.N
    tables = gt_tables(gt_rbboxes, gt_labels, device)
    gt_offset, k_total = tables[2], sum(tables[3])

    # ---- targets of both stages: assignment per image + one target launch per stage ------------------------------------
    centers = geo.centers.unsqueeze(0).expand(B, N, 3)
    t_init = pointset_targets(centers, geo.valid, gt_rbboxes, gt_labels, cfg.init, gt_rbboxes_ignore, tables=tables)
    # the refine stage's proposals are the init-stage point sets (detached) -- with the reference's quirk (head :378-381)
    # of adding the (y, x) offsets to the (x, y) centres unswapped
    proposals = train_ops.points_from_offsets(pts_preds_init, strides, mode=1)
    t_ref = pointset_targets(proposals, geo.valid, gt_rbboxes, gt_labels, cfg.refine, gt_rbboxes_ignore, tables=tables)
    # the ONE host read of the step's loss: positives per image of both stages
    counts = torch.stack([t_init['counts'], t_ref['counts']], 0).tolist()
    p_init = sum(c[0] for c in counts[0])
    p_ref_img = [c[0] for c in counts[1]]
    p_ref = sum(p_ref_img)

    if record is not None:
        lv = lambda x: images_to_levels(x, geo.num_level)                                # noqa: E731
        record['init_target'] = (lv(t_init['labels']), lv(t_init['label_weights']), lv(t_init['rbbox_gt']), None,
                                 lv(t_init['proposal_weights']), sum(max(c[0], 1) for c in counts[0]),
                                 sum(max(c[1], 1) for c in counts[0]), lv(t_init['gt_inds']))

    # ---- init stage losses: all levels at once, one mean per level -------------------------------------------------------
    idx_init = torch.nonzero_static(t_init['proposal_weights'].view(-1) > 0, size=p_init).view(-1)
    lvl_init = geo.level[idx_init % N].long()
    norm_init = (head.point_base_scale * geo.stride[idx_init % N])[:, None]
    pts_init_pos = train_ops.gather_levels(pts_preds_init, strides, idx_init, mode=1) / norm_init
    gt_init_pos = t_init['rbbox_gt'].view(-1, 8)[idx_init] / norm_init
    ones_init = pts_init_pos.new_ones((p_init,))
    per_level = torch.bincount(lvl_init, minlength=num_level)
    loss_rbox_init = _SegmentGIoULoss.apply(pts_init_pos, gt_init_pos, ones_init, lvl_init, num_level, per_level,
                                            head.loss_rbox_init.loss_weight)
    if head.loss_spatial_init is not None:
        loss_border_init = _segment_border_loss(pts_init_pos, gt_init_pos, ones_init, lvl_init, num_level, per_level,
                                                head.loss_spatial_init.loss_weight)
    else:
        loss_border_init = loss_rbox_init.new_zeros((num_level,))

    # ---- refine stage: the positives of all images ----------------------------------------------------------------------------
    labels = t_ref['labels'].view(-1)
    label_weights = t_ref['label_weights'].view(-1)
    rbox_weights = t_ref['proposal_weights'].view(-1)
    idx_ref = torch.nonzero_static(labels > 0, size=p_ref).view(-1)                    # ascending: image-major
    loc_ref = idx_ref % N
    img_ref = torch.div(idx_ref, N, rounding_mode='floor')
    lvl_ref = geo.level[loc_ref]
    gt_ref_pos = t_ref['rbbox_gt'].view(-1, 8)[idx_ref]
    gt_local = t_ref['gt_inds'].view(-1)[idx_ref]                                       # 1-based inside the image
    if record is not None:
        split = lambda x: list(torch.split(x, p_ref_img))                                # noqa: E731
        record['refine_target'] = [list(t_ref['labels'].clone().unbind(0)), list(t_ref['label_weights'].clone().unbind(0)),
                                   list(t_ref['rbbox_gt'].unbind(0)), None, list(t_ref['proposal_weights'].clone().unbind(0)),
                                   split(loc_ref), split(gt_local)]

    # ---- APAA: quality assessment + sample selection (no gradient), all images in one pass --------------------------------
    with torch.no_grad():
        if p_ref > 0:
            feats = [f.detach() for f in base_features]
            qua = _quality(head, feats, img_ref.to(torch.int32), lvl_ref,
                           train_ops.gather_levels([c.detach() for c in cls_scores], strides, idx_ref, mode=0),
                           train_ops.gather_levels([p.detach() for p in pts_preds_init], strides, idx_ref, mode=1),
                           train_ops.gather_levels([p.detach() for p in pts_preds_refine], strides, idx_ref, mode=1),
                           labels[idx_ref], gt_ref_pos, label_weights[idx_ref], rbox_weights[idx_ref])
            gt_global = gt_local + gt_offset.long()[img_ref]                             # a gt's positives all lie in one image
            keep = apaa.apaa_select(qua, gt_global, lvl_ref, k_total, num_level, 6, head.top_ratio)
            keep_f = keep.to(rbox_weights.dtype)
            labels[idx_ref] = labels[idx_ref] * keep.to(labels.dtype)                    # dropped positives become background
            rbox_weights[idx_ref] = keep_f
            num_pos = keep_f.sum()
        else:
            qua = rbox_weights.new_zeros((0,))
            keep = torch.zeros((0,), dtype=torch.bool, device=device)
            keep_f = rbox_weights.new_zeros((0,))
            num_pos = rbox_weights.new_zeros(())
        norm_ref = (head.point_base_scale * geo.stride[loc_ref])
    if record is not None:
        record['qa'] = list(torch.split(qua, p_ref_img))
        record['sel'] = []
        for i, (k_i, n_i) in enumerate(zip(torch.split(keep, p_ref_img), torch.split(norm_ref, p_ref_img))):
            record['sel'].append((labels.view(B, N)[i].clone(), label_weights.view(B, N)[i].clone(),
                                  rbox_weights.view(B, N)[i].clone(), int(k_i.sum()), n_i[k_i].clone()))

    # ---- the classification and refine-stage losses ------------------------------------------------------------------------------
    cls_all = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, C) for c in cls_scores], 1).reshape(-1, C)
    has_pos = (num_pos > 0).to(cls_all.dtype)
    denom = num_pos.clamp(min=1.0)
    # no kept positive at all: the reference returns zero-valued losses (head :455-458)
    losses_cls = head.loss_cls(cls_all, labels, label_weights, avg_factor=denom) * has_pos
    zero = torch.zeros((p_ref,), dtype=torch.long, device=device)
    pts_ref_pos = train_ops.gather_levels(pts_preds_refine, strides, idx_ref, mode=1) / norm_ref[:, None]
    gt_ref_norm = gt_ref_pos / norm_ref[:, None]
    losses_rbox_refine = _SegmentGIoULoss.apply(pts_ref_pos, gt_ref_norm, keep_f, zero, 1, denom.reshape(1),
                                                head.loss_rbox_refine.loss_weight)[0]
    if head.loss_spatial_refine is not None:
        loss_border_refine = _segment_border_loss(pts_ref_pos, gt_ref_norm, keep_f, zero, 1, num_pos.reshape(1),
                                                  head.loss_spatial_refine.loss_weight)
    else:
        loss_border_refine = losses_rbox_refine.new_zeros(1)
    return {
        'loss_cls': losses_cls,
        'loss_rbox_init': list(loss_rbox_init.unbind(0)),
        'loss_rbox_refine': losses_rbox_refine,
        'loss_spatial_init': [v.reshape(1) for v in loss_border_init.unbind(0)],
        'loss_spatial_refine': loss_border_refine,
    }
