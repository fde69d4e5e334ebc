#!/usr/bin/env python3
"""Golden vectors from the reference's own PYTHON code (host logic of the hot path), generated in the build container.

The reference's mmdet package cannot be imported as a whole (mmcv, CUDA extensions), so the few source files that hold
the host logic are loaded individually from /root/reference under stub parents; native ops they call are routed to
the CPU oracle (itself pinned against the reference's C/CUDA sources).  Nothing from the reference is copied: the files
are executed where they lie.  Output: tests/golden/apaa_py.npz

    python tests/golden/make_golden_py.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

from oracle import orp_oracle as O  # noqa: E402
from orientedreppoints_amd import synthetic as S  # noqa: E402


def stub(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    m.__dict__.update(attrs)
    return m


def load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    m.__package__ = modname.rsplit('.', 1)[0]
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def t2n(t):
    return t.detach().cpu().numpy()


# ---- oracle-backed stand-ins for the native ops -------------------------------------------------------------------
def _convex_overlaps(gt, points):
    return torch.from_numpy(O.convex_iou(t2n(points), t2n(gt)).T.copy())


def _minaerarect(p):
    return torch.from_numpy(O.minarearect(t2n(p)))


def _chamfer(p1, p2, distance_weight=0.05, eps=1e-12, use_cuda=True):
    d1, d2, _, _ = O.chamfer_forward(t2n(p1), t2n(p2))
    d1, d2 = torch.from_numpy(d1), torch.from_numpy(d2)
    d1 = torch.sqrt(torch.clamp(d1, eps)); d2 = torch.sqrt(torch.clamp(d2, eps))
    return (d1.mean(-1) + d2.mean(-1)) / 2.0 * distance_weight


class _FocalNone(object):
    loss_weight = 1.0

    def __call__(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        loss = torch.from_numpy(O.focal_forward(t2n(pred), t2n(target), 2.0, 0.25))
        if weight is not None:
            loss = loss * weight.view(-1, 1)
        assert reduction_override == 'none'
        return self.loss_weight * loss


class _GIoUNone(object):
    loss_weight = 1.0

    def __call__(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        out = torch.from_numpy(O.convex_giou(t2n(pred), t2n(target)))
        loss = 1 - out[:, 18]
        if weight is not None:
            loss = loss * weight
        return self.loss_weight * loss


def main():
    torch.Tensor.cuda = lambda self, *a, **k: self     # AssignResult moves tensors with .cuda(); no GPU here
    class NiceRepr(object):
        pass
    stub('mmdet'); stub('mmdet.utils', util_mixins=types.SimpleNamespace(NiceRepr=NiceRepr), print_log=print)
    stub('mmdet.utils.util_mixins', NiceRepr=NiceRepr)
    stub('mmdet.ops'); stub('mmdet.ops.iou', convex_overlaps=_convex_overlaps)
    stub('mmdet.core'); stub('mmdet.core.bbox'); stub('mmdet.core.bbox.assigners')
    load('mmdet.core.bbox.assigners.assign_result', 'mmdet/core/bbox/assigners/assign_result.py')
    load('mmdet.core.bbox.assigners.base_assigner', 'mmdet/core/bbox/assigners/base_assigner.py')
    pa = load('mmdet.core.bbox.assigners.point_assigner', 'mmdet/core/bbox/assigners/point_assigner.py')
    mia = load('mmdet.core.bbox.assigners.max_iou_assigner', 'mmdet/core/bbox/assigners/max_iou_assigner.py')

    g = {}
    # ---- points of one 256x256 image (strides 8..128): [N,3] ----------------------------------------------------
    def grid_points(size=256):
        pts = []
        for s in (8, 16, 32, 64, 128):
            f = size // s
            ys, xs = np.meshgrid(np.arange(f) * s, np.arange(f) * s, indexing='ij')
            pts.append(np.stack([xs.ravel(), ys.ravel(), np.full(f * f, s)], 1))
        return np.concatenate(pts, 0).astype(np.float32)
    points = grid_points()
    g['points'] = points
    rng = np.random.RandomState(0)
    gts = (S.gen_polys(23, 5, wh=(6, 200))[:, :8] / 4.0).astype(np.float32)       # inside 256x256, all sizes
    labels = rng.randint(1, 16, size=23).astype(np.int64)
    g['gts'] = gts; g['gt_labels'] = labels
    for pos_num in (1, 3):
        r = pa.PointAssigner(scale=4, pos_num=pos_num).assign(torch.from_numpy(points), torch.from_numpy(gts), None,
                                                              torch.from_numpy(labels))
        g['pa_gt_inds_%d' % pos_num] = t2n(r.gt_inds); g['pa_labels_%d' % pos_num] = t2n(r.labels)

    # ---- MaxIoUAssigner on real convex overlaps ------------------------------------------------------------------
    N = points.shape[0]
    ctr = points[:, :2] + points[:, 2:3] / 2
    psets = S.gen_pointsets(N, 7, around=ctr.astype(np.float64)).astype(np.float32)
    # make a third of the point sets sit on gts so that positives exist
    idx = rng.choice(N, N // 3, replace=False)
    gsel = rng.randint(0, 23, size=idx.size)
    gc = gts.reshape(-1, 4, 2).mean(1)[gsel]
    psets[idx] = S.gen_pointsets(idx.size, 8, around=gc.astype(np.float64) + rng.normal(0, 3, (idx.size, 2))).astype(np.float32)
    g['psets'] = psets
    ov = O.convex_iou(psets, gts).T.copy()          # [K, N]
    g['overlaps'] = ov
    a = mia.MaxIoUAssigner(pos_iou_thr=0.1, neg_iou_thr=0.1, min_pos_iou=0, ignore_iof_thr=-1)
    r = a.assign_wrt_overlaps(torch.from_numpy(ov.copy()), torch.from_numpy(labels))
    g['mia_gt_inds'] = t2n(r.gt_inds); g['mia_max_overlaps'] = t2n(r.max_overlaps); g['mia_labels'] = t2n(r.labels)
    a2 = mia.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.3, min_pos_iou=0.2, ignore_iof_thr=-1)
    r2 = a2.assign_wrt_overlaps(torch.from_numpy(ov.copy()), torch.from_numpy(labels))
    g['mia2_gt_inds'] = t2n(r2.gt_inds)

    # ---- head methods (unbound, fake self) ------------------------------------------------------------------------
    HEADS = types.SimpleNamespace(register_module=lambda cls: cls)
    stub('mmcv'); stub('mmcv.cnn', normal_init=None, constant_init=None)
    stub('mmdet.core', PointGenerator=None, multi_apply=None, multiclass_rnms=None, levels_to_images=None)
    stub('mmdet.ops', ConvModule=None, DeformConv=None)
    stub('mmdet.models'); stub('mmdet.models.anchor_heads')
    stub('mmdet.models.builder', build_loss=None); stub('mmdet.models.registry', HEADS=HEADS)
    stub('mmdet.models.utils', bias_init_with_prob=None)
    stub('mmdet.core.bbox', init_pointset_target=None, refine_pointset_target=None)
    stub('mmdet.ops.minarearect', minaerarect=_minaerarect)
    stub('mmdet.ops.chamfer_distance', ChamferDistance2D=_chamfer)
    hm = load('mmdet.models.anchor_heads.orientedreppoints_head', 'mmdet/models/anchor_heads/orientedreppoints_head.py')
    H = hm.OrientedRepPointsHead
    fake = types.SimpleNamespace(num_points=9, top_ratio=0.4, point_base_scale=2, point_strides=[8, 16, 32, 64, 128],
                                 loss_cls=_FocalNone(), loss_rbox_refine=_GIoUNone())
    fake.sampling_points = lambda c, n: H.sampling_points(fake, c, n)
    fake.feature_cosine_similarity = lambda f: H.feature_cosine_similarity(fake, f)

    corners = torch.from_numpy(gts)
    g['sampling_points'] = t2n(H.sampling_points(fake, corners, 10))
    feats = torch.from_numpy(rng.normal(size=(50, 9, 256)).astype(np.float32))
    feats[:5] *= 1e-3                                 # exercise the 1e-2 norm clamp
    g['cos_feats'] = t2n(feats); g['cos_out'] = t2n(H.feature_cosine_similarity(fake, feats))

    # get_adaptive_points_feature (F.grid_sample, align_corners default of the running torch = False)
    fmap = torch.from_numpy(rng.normal(size=(2, 16, 8, 8)).astype(np.float32))
    locs = torch.from_numpy(rng.uniform(-10, 70, size=(2, 64, 18)).astype(np.float32))
    g['gapf_feat'] = t2n(fmap); g['gapf_locs'] = t2n(locs)
    g['gapf_out'] = t2n(H.get_adaptive_points_feature(fake, fmap, locs, 8)[0])

    # points_quality_assessment + point_samples_selection on the refine-stage assignment above
    gt_inds = r.gt_inds
    pos_inds = (gt_inds > 0).nonzero().view(-1)
    P = pos_inds.numel()
    cls_score = torch.from_numpy(rng.normal(-2, 1.5, size=(N, 15)).astype(np.float32))
    pts_init = torch.from_numpy(psets)
    pts_refine = torch.from_numpy(psets + rng.normal(0, 1.0, psets.shape).astype(np.float32))
    pfeat = torch.from_numpy(rng.normal(size=(N, 9, 32)).astype(np.float32))
    label = r.labels.clone()
    rbbox_gt = torch.zeros(N, 8); rbbox_gt[pos_inds] = torch.from_numpy(gts)[gt_inds[pos_inds] - 1]
    label_weight = torch.ones(N); rbox_weight = torch.zeros(N); rbox_weight[pos_inds] = 1.0
    qua, = H.points_quality_assessment(fake, pfeat, cls_score, pts_init, pts_refine, label, rbbox_gt, label_weight,
                                       rbox_weight, pos_inds)
    g['qa_cls_score'] = t2n(cls_score); g['qa_pts_refine'] = t2n(pts_refine); g['qa_pfeat'] = t2n(pfeat)
    g['qa_pos_inds'] = t2n(pos_inds); g['qa_rbbox_gt'] = t2n(rbbox_gt); g['qa_out'] = t2n(qua)
    nlev = [1024, 256, 64, 16, 4]
    lab2, lw2, rw2, num_pos, pnt = H.point_samples_selection(fake, qua, label.clone(), label_weight.clone(),
                                                             rbox_weight.clone(), pos_inds, gt_inds[pos_inds],
                                                             num_proposals_each_level=nlev, num_level=5)
    g['sel_label'] = t2n(lab2); g['sel_label_weight'] = t2n(lw2); g['sel_rbox_weight'] = t2n(rw2)
    g['sel_num_pos'] = np.array(num_pos); g['sel_pos_normalize_term'] = t2n(pnt)
    g['sel_pos_gt_inds'] = t2n(gt_inds[pos_inds])
    print('P =', P, 'num_pos after selection =', num_pos)

    # offset_to_pts
    centers = [[torch.from_numpy(points[:1024]), torch.from_numpy(points[1024:1280])]]
    preds = [torch.from_numpy(rng.normal(size=(1, 18, 32, 32)).astype(np.float32)),
             torch.from_numpy(rng.normal(size=(1, 18, 16, 16)).astype(np.float32))]
    fake2 = types.SimpleNamespace(num_points=9, point_strides=[8, 16])
    otp = H.offset_to_pts(fake2, centers, preds)
    g['otp_pred0'] = t2n(preds[0]); g['otp_pred1'] = t2n(preds[1]); g['otp_out0'] = t2n(otp[0]); g['otp_out1'] = t2n(otp[1])
    np.savez_compressed(os.path.join(OUT, 'apaa_py.npz'), **g)
    print('written', os.path.join(OUT, 'apaa_py.npz'))


if __name__ == '__main__':
    main()
