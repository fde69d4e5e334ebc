"""init_pointset_target / refine_pointset_target: the per-stage training targets of the head.

Interface of mmdet/core/bbox/pointset_target.py:6-230 (same arguments, same result tuples); the implementation is this
package's own: every image is assigned by the device assigners and ONE `orp_pointset_target` launch
(mmdet_ops/train_ops.py, csrc/orp_train.hip) writes labels, label weights, the gt box of every positive, proposal weights
and gt indices for all images at their full-N positions -- there is no per-image target function, no `unmap`, and the
per-level views the init stage returns are slices of the batched tensors, not copies.  `pointset_targets` is the batched
core the head's loss() calls directly (it also returns the positive / negative counts as a device tensor, so the caller
decides when to pay for a host read)."""
import torch

from ..mmdet_ops import train_ops
from .registry import build_assigner


def images_to_levels(target, num_level_anchors):
    """[B, N, ...] (or a list of B [N, ...] tensors) -> per-level slices [B, n_l, ...] (squeezed for B = 1, as the
    reference's stack + squeeze(0) does)."""
    if isinstance(target, (list, tuple)):
        target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_level_anchors:
        out.append(target[:, start:start + n].squeeze(0))
        start += n
    return out


def assign_images(assigner, proposals, valid, gt_rbboxes_list, gt_labels_list, gt_rbboxes_ignore_list=None):
    """gt_inds [B, N] int64 of all images (0 at invalid locations).  proposals [B, N, D]; valid [B, N] bool or None (every
    location valid: no compaction, no host synchronisation)."""
    B, N = proposals.shape[:2]
    rows = []
    for b in range(B):
        ign = gt_rbboxes_ignore_list[b] if gt_rbboxes_ignore_list is not None else None
        lab = None                   # the assigners' own label lookup is not needed: orp_pointset_target reads the labels
        if valid is None:
            rows.append(assigner.assign(proposals[b], gt_rbboxes_list[b], ign, lab).gt_inds)
        else:
            inside = valid[b]
            gi = proposals.new_zeros((N,), dtype=torch.long)
            gi[inside] = assigner.assign(proposals[b][inside], gt_rbboxes_list[b], ign, lab).gt_inds
            rows.append(gi)
    return torch.stack(rows, 0)


def gt_tables(gt_rbboxes_list, gt_labels_list, device):
    """(gt boxes [K,8] and labels [K] of all images concatenated, gt_offset [B+1] int32 on the device, K per image)."""
    counts = [int(g.shape[0]) for g in gt_rbboxes_list]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    boxes = torch.cat([g.reshape(-1, 8) for g in gt_rbboxes_list], 0) if sum(counts) else \
        torch.zeros((0, 8), dtype=torch.float32, device=device)
    labels = None
    if gt_labels_list is not None and all(l is not None for l in gt_labels_list):
        labels = torch.cat([l.reshape(-1) for l in gt_labels_list], 0) if sum(counts) else \
            torch.zeros((0,), dtype=torch.long, device=device)
    return boxes, labels, torch.tensor(offs, dtype=torch.int32, device=device), counts


def pointset_targets(proposals, valid, gt_rbboxes_list, gt_labels_list, cfg, gt_rbboxes_ignore_list=None, tables=None,
                     want_proposals=False):
    """The batched core: assignment of every image + one target launch.  proposals [B, N, D]; valid [B, N] bool or None.
    Returns the dict of `train_ops.pointset_target` (labels, label_weights, rbbox_gt, proposal_weights, gt_inds,
    counts [B,2] int32: positives / negatives per image, and pos_proposals when asked for)."""
    assigner = build_assigner(cfg.assigner)
    gt_inds = assign_images(assigner, proposals, valid, gt_rbboxes_list, gt_labels_list, gt_rbboxes_ignore_list)
    boxes, labels, offs, _ = tables if tables is not None else gt_tables(gt_rbboxes_list, gt_labels_list, proposals.device)
    return train_ops.pointset_target(gt_inds, valid, boxes, labels, offs, pos_weight=cfg.pos_weight,
                                     proposals=proposals if want_proposals else None)


def _batched(proposals_list, valid_flag_list, num_imgs):
    """per image lists of per-level tensors -> ([B,N,D] proposals, [B,N] valid or None when every flag is set)."""
    props = torch.stack([torch.cat(list(p)) if isinstance(p, (list, tuple)) else p for p in proposals_list], 0)
    flags = torch.stack([torch.cat(list(v)) if isinstance(v, (list, tuple)) else v for v in valid_flag_list], 0).bool()
    assert props.shape[:2] == flags.shape and props.size(0) == num_imgs
    return props, (None if bool(flags.all()) else flags)


def init_pointset_target(proposals_list, valid_flag_list, gt_rbboxes_list, img_metas, cfg,
                         gt_rbboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                         unmap_outputs=True):
    """-> (labels, label_weights, rbbox_gt, pos_proposals, proposal_weights per LEVEL ([B, n_l, ...] slices),
    num_total_pos, num_total_neg, gt_inds per level); None when an image has no valid location."""
    assert not sampling, 'the focal-loss configs use PseudoSampler (sampling=False)'
    num_imgs = len(img_metas)
    assert len(proposals_list) == len(valid_flag_list) == num_imgs
    num_level_proposals = [p.size(0) for p in proposals_list[0]]
    props, valid = _batched(proposals_list, valid_flag_list, num_imgs)
    if valid is not None and not bool(valid.any(dim=1).all()):
        return None
    t = pointset_targets(props, valid, gt_rbboxes_list, gt_labels_list, cfg, gt_rbboxes_ignore_list, want_proposals=True)
    counts = t['counts'].tolist()
    num_total_pos = sum(max(c[0], 1) for c in counts)
    num_total_neg = sum(max(c[1], 1) for c in counts)
    lv = lambda x: images_to_levels(x, num_level_proposals)                                  # noqa: E731
    return (lv(t['labels']), lv(t['label_weights']), lv(t['rbbox_gt']), lv(t['pos_proposals']), lv(t['proposal_weights']),
            num_total_pos, num_total_neg, lv(t['gt_inds']))


def refine_pointset_target(proposals_list, valid_flag_list, gt_rbboxes_list, img_metas, cfg,
                           gt_rbboxes_ignore_list=None, gt_labels_list=None, label_channels=1, sampling=True,
                           unmap_outputs=True):
    """-> per IMAGE lists (labels, label_weights, rbbox_gt, pos_proposals, proposal_weights, pos_inds, pos_gt_index)."""
    assert not sampling, 'the focal-loss configs use PseudoSampler (sampling=False)'
    num_imgs = len(img_metas)
    assert len(proposals_list) == len(valid_flag_list) == num_imgs
    props, valid = _batched(proposals_list, valid_flag_list, num_imgs)
    t = pointset_targets(props, valid, gt_rbboxes_list, gt_labels_list, cfg, gt_rbboxes_ignore_list, want_proposals=True)
    pos_inds, pos_gt_index = [], []
    for b in range(num_imgs):
        idx = (t['labels'][b] > 0).nonzero().view(-1)
        pos_inds.append(idx)
        pos_gt_index.append(t['gt_inds'][b][idx])
    per_image = lambda x: list(x.unbind(0))                                                   # noqa: E731
    return (per_image(t['labels']), per_image(t['label_weights']), per_image(t['rbbox_gt']), per_image(t['pos_proposals']),
            per_image(t['proposal_weights']), pos_inds, pos_gt_index)
