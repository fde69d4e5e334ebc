"""ctypes front-end for the CPU oracle (oracle/orp_oracle.c) and, when present, oracle/_ref/libref_orp.so.

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (orientedreppoints_amd/) must never import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, f) for f in ("orp_oracle.c", "orp_oracle2.c", "orp_oracle3.c", "orp_oracle4.c", "orp_polyclip.inc", "orp_hull.inc")]
LIB = os.path.join(HERE, "liborp_oracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libref_orp.so")

_c_f32p = ctypes.c_void_p
_lib = None
_ref = None


def build(force=False):
    """gcc -O2 -ffp-contract=off the restatement into oracle/liborp_oracle.so (a few seconds)."""
    srcs = [s for s in SRC if os.path.exists(s)]
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs)):
        return LIB
    cfiles = [s for s in srcs if s.endswith(".c")]
    cmd = ["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-Wno-parentheses", "-Wno-misleading-indentation",
           "-Wno-alloc-size-larger-than"] + cfiles + ["-o", LIB, "-lm"]
    subprocess.check_call(cmd)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB)
        L.orc_quad_iou_f32.restype = ctypes.c_float
        L.orc_poly_nms_iou_f32.restype = ctypes.c_float
        L.orc_polyiou_f64.restype = ctypes.c_double
        _lib = L
    return _lib


def ref():
    """The reference's own functions (None when oracle/_ref was never built, e.g. no /root/reference)."""
    global _ref
    if _ref is None and os.path.exists(REF_LIB):
        L = ctypes.CDLL(REF_LIB)
        for n in ("ref_rnms_iou", "ref_rnms_cpu_iou", "ref_poly_nms_iou", "ref_poly_overlaps_iou", "ref_convex_iou"):
            getattr(L, n).restype = ctypes.c_float
        L.ref_polyiou_iou.restype = ctypes.c_double
        _ref = L
    return _ref


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---------------------------------------------------------------------------------------------------------
# oracle (restatement)
# ---------------------------------------------------------------------------------------------------------
def quad_iou_matrix(a, b, guard=False):
    """fp32 quad IoU, all pairs; rows may carry extra columns (e.g. [x1..y4, score])."""
    a, b = _f32(a), _f32(b)
    assert a.shape[1] == b.shape[1]
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().orc_quad_iou_matrix_f32(_p(a), a.shape[0], _p(b), b.shape[0], a.shape[1], int(guard), _p(out))
    return out


def polyiou(p, q):
    p, q = _f64(p), _f64(q)
    return lib().orc_polyiou_f64(_p(p), _p(q))


def sort_order(scores):
    s = _f32(scores)
    order = np.empty(s.shape[0], np.int32)
    lib().orc_sort_order(_p(s), s.shape[0], 1, _p(order))
    return order


def nms_sorted(dets, thr, guard=False):
    d = _f32(dets)
    keep = np.empty(d.shape[0], np.int32)
    n = lib().orc_nms_sorted_f32(_p(d), d.shape[0], d.shape[1], ctypes.c_float(thr), int(guard), _p(keep))
    return keep[:n].copy()


def rnms(dets, thr):
    """rnms_cuda semantics: original indices of kept boxes, ascending."""
    d = _f32(dets)
    keep = np.empty(d.shape[0], np.int64)
    n = lib().orc_rnms(_p(d), d.shape[0], ctypes.c_float(thr), _p(keep))
    return keep[:n].copy()


def poly_gpu_nms(dets, thr):
    """poly_nms.pyx:9-24 semantics: numpy argsort()[::-1] order, fp32 devPolyIoU, returns kept ORIGINAL indices
    in score order."""
    d = _f32(dets)
    order = d[:, 8].argsort()[::-1]
    keep = nms_sorted(d[order], thr, guard=True)
    return [int(x) for x in order[keep]]


def py_cpu_nms_poly(dets, thr):
    """DOTA_devkit/ResultMerge.py:18-41 (fp64), same visiting order as the reference (numpy argsort()[::-1])."""
    d = _f64(dets)
    order = np.ascontiguousarray(d[:, 8].argsort()[::-1], dtype=np.int64)
    keep = np.empty(d.shape[0], np.int64)
    n = lib().orc_py_cpu_nms_poly(_p(d), d.shape[0], _p(order), ctypes.c_double(thr), _p(keep))
    return [int(x) for x in keep[:n]]


def py_cpu_nms_poly_fast(dets, thr, use_ref=False):
    """ResultMerge_multi_process.py:60-121: polyiou only where the HBBs overlap.  dets [n,9] fp64 -> kept indices."""
    d = _f64(dets)
    order = np.ascontiguousarray(d[:, 8].argsort()[::-1], np.int64)
    keep = np.empty(d.shape[0], np.int64)
    fn = ref().ref_py_cpu_nms_poly_fast if use_ref else lib().orc_py_cpu_nms_poly_fast
    n = fn(_p(d), d.shape[0], _p(order), ctypes.c_double(thr), _p(keep))
    return keep[:n].copy()


def ref_py_cpu_nms_poly(dets, thr):
    """The greedy loop around the REFERENCE's own polyiou.cpp iou_poly (oracle/_ref)."""
    d = _f64(dets)
    order = np.ascontiguousarray(d[:, 8].argsort()[::-1], np.int64)
    keep = np.empty(d.shape[0], np.int64)
    n = ref().ref_py_cpu_nms_poly(_p(d), d.shape[0], _p(order), ctypes.c_double(thr), _p(keep))
    return keep[:n].copy()


def ref_rnms_cpu_hard(dets_sorted, thr):
    """Hard NMS with rnms_cpu.cpp's own fp32 rotate_iou over score-sorted dets [n,9] -> kept positions."""
    d = _f32(dets_sorted)
    keep = np.empty(d.shape[0], np.int32)
    n = ref().ref_rnms_cpu_hard(_p(d), d.shape[0], ctypes.c_float(thr), _p(keep))
    return keep[:n].copy()


def ref_polyiou_many(a, b):
    a, b = _f64(a), _f64(b)
    out = np.empty(a.shape[0], np.float64)
    ref().ref_polyiou_many(_p(a), _p(b), a.shape[0], _p(out))
    return out


def voc_best_match(dets, det_img, gts, gt_off):
    """dota_evaluation_task1.py:160-206 per detection (numpy fp64 HBB pre-filter with the "+ 1." convention, then
    polyiou.iou_poly(GT, detection) on the survivors, np.max / np.argmax): (ovmax [nd], jmax [nd]); (-inf, -1) when no
    ground truth of the image passes the pre-filter.  Plain python loop: small cases only."""
    dets = _f64(dets); gts = _f64(gts)
    ovmax = np.full(len(dets), -np.inf); jmax = np.full(len(dets), -1, np.int32)
    for d in range(len(dets)):
        bb = dets[d]
        BBGT = gts[gt_off[det_img[d]]:gt_off[det_img[d] + 1]]
        if BBGT.size == 0:
            continue
        BBGT_xmin = np.min(BBGT[:, 0::2], axis=1); BBGT_ymin = np.min(BBGT[:, 1::2], axis=1)
        BBGT_xmax = np.max(BBGT[:, 0::2], axis=1); BBGT_ymax = np.max(BBGT[:, 1::2], axis=1)
        bb_xmin = np.min(bb[0::2]); bb_ymin = np.min(bb[1::2]); bb_xmax = np.max(bb[0::2]); bb_ymax = np.max(bb[1::2])
        ixmin = np.maximum(BBGT_xmin, bb_xmin); iymin = np.maximum(BBGT_ymin, bb_ymin)
        ixmax = np.minimum(BBGT_xmax, bb_xmax); iymax = np.minimum(BBGT_ymax, bb_ymax)
        iw = np.maximum(ixmax - ixmin + 1., 0.); ih = np.maximum(iymax - iymin + 1., 0.)
        inters = iw * ih
        uni = ((bb_xmax - bb_xmin + 1.) * (bb_ymax - bb_ymin + 1.) +
               (BBGT_xmax - BBGT_xmin + 1.) * (BBGT_ymax - BBGT_ymin + 1.) - inters)
        with np.errstate(all='ignore'):
            overlaps = inters / uni
        keep = np.where(overlaps > 0)[0]
        if len(keep) > 0:
            ov = [polyiou(BBGT[j], bb) for j in keep]
            ovmax[d] = np.max(ov)
            jmax[d] = keep[int(np.argmax(ov))]
    return ovmax, jmax


def poly_overlaps(boxes, query):
    b, q = _f32(boxes), _f32(query)
    out = np.empty((b.shape[0], q.shape[0]), np.float32)
    lib().orc_poly_overlaps(_p(b), b.shape[0], _p(q), q.shape[0], _p(out))
    return out


def rotbox2poly(boxes):
    b = _f32(boxes)
    out = np.empty((b.shape[0], 8), np.float32)
    lib().orc_rotbox2poly(_p(b), b.shape[0], _p(out))
    return out


def minarearect(pts):
    p = _f32(pts)
    out = np.empty((p.shape[0], 8), np.float32)
    lib().orc_minarearect(_p(p), p.shape[0], _p(out))
    return out


def minarearect_margin(pts):
    """Relative area gap between the best and the second-best candidate rectangle (tie diagnostics for the tests)."""
    p = _f32(pts)
    out = np.empty((p.shape[0],), np.float32)
    lib().orc_minarearect_margin(_p(p), p.shape[0], _p(out))
    return out


def convex_iou(pts, gts):
    p, g = _f32(pts), _f32(gts)
    out = np.empty((p.shape[0], g.shape[0]), np.float32)
    lib().orc_convex_iou(_p(p), p.shape[0], _p(g), g.shape[0], _p(out))
    return out


def stats():
    L = lib()
    return dict(max_clip_n=L.orc_stat_max_clip_n(), clip_overflow=L.orc_stat_clip_overflow())


def stats_reset():
    lib().orc_stat_reset()


# ---------------------------------------------------------------------------------------------------------
# reference (_ref) -- same call shapes, for pinning the restatement
# ---------------------------------------------------------------------------------------------------------
def ref_quad_iou_matrix(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    ref().ref_rnms_iou_matrix(_p(a), a.shape[0], _p(b), b.shape[0], a.shape[1], _p(out))
    return out


def ref_pair_iou(fn, p, q):
    p, q = _f32(p), _f32(q)
    return getattr(ref(), fn)(_p(p), _p(q))


def ref_polyiou(p, q):
    p, q = _f64(p), _f64(q)
    return ref().ref_polyiou_iou(_p(p), _p(q))


def ref_nms_sorted(dets, thr, which=0):
    d = _f32(dets)
    assert d.shape[1] == 9
    keep = np.empty(d.shape[0], np.int32)
    n = ref().ref_nms_sorted(_p(d), d.shape[0], ctypes.c_float(thr), which, _p(keep))
    return keep[:n].copy()


def ref_poly_overlaps(boxes, query):
    b, q = _f32(boxes), _f32(query)
    out = np.empty((b.shape[0], q.shape[0]), np.float32)
    ref().ref_poly_overlaps(_p(b), b.shape[0], _p(q), q.shape[0], _p(out))
    return out


def ref_minarearect(pts):
    p = _f32(pts)
    out = np.empty((p.shape[0], 8), np.float32)
    ref().ref_minarearect_batch(_p(p), p.shape[0], _p(out))
    return out


def ref_convex_iou(pts, gts):
    p, g = _f32(pts), _f32(gts)
    out = np.empty((p.shape[0], g.shape[0]), np.float32)
    ref().ref_convex_iou_matrix(_p(p), p.shape[0], _p(g), g.shape[0], _p(out))
    return out


def ref_convex_giou(pts, gts):
    p, g = _f32(pts), _f32(gts)
    out = np.empty((p.shape[0], 19), np.float32)
    ref().ref_convex_giou_batch(_p(p), _p(g), p.shape[0], _p(out))
    return out


# ---------------------------------------------------------------------------------------------------------
# part 2: pointsJf / chamfer / focal
# ---------------------------------------------------------------------------------------------------------
def points_justify(points, polys):
    p, q = _f32(points), _f32(polys)
    out = np.empty((p.shape[0], q.shape[0]), np.float32)
    lib().orc_points_justify(_p(p), p.shape[0], _p(q), q.shape[0], _p(out))
    return out


def points_in_quad_aligned(pts18, quads):
    p, q = _f32(pts18), _f32(quads)
    out = np.empty((p.shape[0], 9), np.float32)
    lib().orc_points_in_quad_aligned(_p(p), _p(q), p.shape[0], _p(out))
    return out


def chamfer_forward(xyz1, xyz2):
    a, b = _f32(xyz1), _f32(xyz2)
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    d1 = np.empty((B, n), np.float32); i1 = np.empty((B, n), np.int32)
    d2 = np.empty((B, m), np.float32); i2 = np.empty((B, m), np.int32)
    lib().orc_chamfer_nn(_p(a), _p(b), B, n, m, _p(d1), _p(i1))
    lib().orc_chamfer_nn(_p(b), _p(a), B, m, n, _p(d2), _p(i2))
    return d1, d2, i1, i2


def chamfer_backward(xyz1, xyz2, g1, g2, i1, i2):
    a, b = _f32(xyz1), _f32(xyz2)
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    ga = np.zeros_like(a); gb = np.zeros_like(b)
    g1, g2 = _f32(g1), _f32(g2)
    i1 = np.ascontiguousarray(i1, np.int32); i2 = np.ascontiguousarray(i2, np.int32)
    lib().orc_chamfer_grad(_p(a), _p(b), B, n, m, _p(g1), _p(i1), _p(ga), _p(gb))
    lib().orc_chamfer_grad(_p(b), _p(a), B, m, n, _p(g2), _p(i2), _p(gb), _p(ga))
    return ga, gb


def focal_forward(logits, targets, gamma, alpha):
    x = _f32(logits); t = np.ascontiguousarray(targets, np.int64)
    out = np.empty_like(x)
    lib().orc_focal_forward(_p(x), _p(t), x.shape[0], x.shape[1], ctypes.c_float(gamma), ctypes.c_float(alpha), _p(out))
    return out


def focal_backward(logits, targets, d_losses, gamma, alpha):
    x = _f32(logits); t = np.ascontiguousarray(targets, np.int64); g = _f32(d_losses)
    out = np.empty_like(x)
    lib().orc_focal_backward(_p(x), _p(t), _p(g), x.shape[0], x.shape[1], ctypes.c_float(gamma), ctypes.c_float(alpha), _p(out))
    return out


def ref_points_justify(points, polys):
    p, q = _f32(points), _f32(polys)
    out = np.full((p.shape[0], q.shape[0]), -1, np.float32)
    ref().ref_points_justify(_p(p), p.shape[0], _p(q), q.shape[0], _p(out))
    return out


def ref_chamfer_nn(xyz, xyz2):
    a, b = _f32(xyz), _f32(xyz2)
    B, n, m = a.shape[0], a.shape[1], b.shape[1]
    d = np.empty((B, n), np.float32); i = np.empty((B, n), np.int32)
    ref().ref_chamfer_nn(B, n, _p(a), m, _p(b), _p(d), _p(i))
    return d, i


def ref_focal_forward(logits, targets, gamma, alpha):
    x = _f32(logits); t = np.ascontiguousarray(targets, np.int64)
    out = np.empty_like(x)
    ref().ref_focal_forward(_p(x), _p(t), x.shape[0], x.shape[1], ctypes.c_float(gamma), ctypes.c_float(alpha), _p(out))
    return out


def ref_focal_backward(logits, targets, d_losses, gamma, alpha):
    x = _f32(logits); t = np.ascontiguousarray(targets, np.int64); g = _f32(d_losses)
    out = np.empty_like(x)
    ref().ref_focal_backward(_p(x), _p(t), _p(g), x.shape[0], x.shape[1], ctypes.c_float(gamma), ctypes.c_float(alpha), _p(out))
    return out


# ---------------------------------------------------------------------------------------------------------
# deformable convolution
# ---------------------------------------------------------------------------------------------------------
def _odim(n, pad, dil, k, stride):
    return (n + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def dcn_im2col(x, offset, kh, kw, pad, stride, dil, dg=1, use_ref=False):
    x, offset = _f32(x), _f32(offset)
    B, C, H, W = x.shape
    Ho, Wo = _odim(H, pad, dil, kh, stride), _odim(W, pad, dil, kw, stride)
    col = np.zeros((C * kh * kw, B, Ho, Wo), np.float32)
    fn = ref().ref_dcn_im2col if use_ref else lib().orc_dcn_im2col
    fn(_p(x), _p(offset), B, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dg, _p(col))
    return col


def dcn_forward(x, offset, weight, stride=1, pad=1, dil=1, groups=1, dg=1, mask=None, bias=None):
    x, offset, weight = _f32(x), _f32(offset), _f32(weight)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = _odim(H, pad, dil, kh, stride), _odim(W, pad, dil, kw, stride)
    out = np.empty((B, Cout, Ho, Wo), np.float32)
    m = _f32(mask) if mask is not None else None
    bb = _f32(bias) if bias is not None else None
    lib().orc_dcn_forward(_p(x), _p(offset), _p(m) if m is not None else None, _p(weight),
                          _p(bb) if bb is not None else None, _p(out), B, C, H, W, Cout, kh, kw, stride, stride, pad,
                          pad, dil, dil, groups, dg)
    return out


def convex_giou(pts, gts, return_flags=False):
    """[P,19] = 18 grads + giou per aligned pair; flags[P] = 1 where the reference itself is undefined (scratch overflow)."""
    p, g = _f32(pts), _f32(gts)
    out = np.empty((p.shape[0], 19), np.float32)
    fl = np.zeros(p.shape[0], np.int32)
    lib().orc_convex_giou(_p(p), _p(g), p.shape[0], _p(out), _p(fl))
    return (out, fl) if return_flags else out


# ---------------------------------------------------------------------------------------------------------
# part 4: assigners / APAA selection
# ---------------------------------------------------------------------------------------------------------
def point_assign(points, gts, scale=4, pos_num=1):
    p, g = _f32(points), _f32(gts).reshape(-1, 8)
    out = np.zeros(p.shape[0], np.int64)
    lib().orc_point_assign(_p(p), p.shape[0], _p(g), g.shape[0], ctypes.c_float(scale), int(pos_num), _p(out))
    return out


def max_iou_assign(overlaps_nk, pos_thr, neg_thr, min_pos_iou=0.0, assign_all=True):
    ov = _f32(overlaps_nk)
    n, k = ov.shape
    lo, hi = (0.0, neg_thr) if not isinstance(neg_thr, tuple) else neg_thr
    gi = np.empty(n, np.int64); mo = np.empty(n, np.float32)
    lib().orc_max_iou_assign(_p(ov), n, k, ctypes.c_float(pos_thr), ctypes.c_float(lo), ctypes.c_float(hi),
                             ctypes.c_float(min_pos_iou), int(assign_all), _p(gi), _p(mo))
    return gi, mo


def sample_points(feat_chw, stride, pts18):
    f, p = _f32(feat_chw), _f32(pts18)
    C, H, W = f.shape
    out = np.empty((p.shape[0], 9, C), np.float32)
    lib().orc_sample_points(_p(f), C, H, W, ctypes.c_float(stride), _p(p), p.shape[0], _p(out))
    return out


def feature_dissimilarity(f):
    f = _f32(f)
    out = np.empty(f.shape[0], np.float32)
    lib().orc_feature_dissimilarity(_p(f), f.shape[0], f.shape[2], _p(out))
    return out


def apaa_select(q, pos_gt, pos_lvl, num_gt, num_level=5, per_level_k=6, top_ratio=0.4):
    q = _f32(q); g = np.ascontiguousarray(pos_gt, np.int64); l = np.ascontiguousarray(pos_lvl, np.int32)
    keep = np.zeros(q.shape[0], np.uint8)
    lib().orc_apaa_select(_p(q), _p(g), _p(l), q.shape[0], int(num_gt), int(num_level), int(per_level_k),
                          ctypes.c_double(top_ratio), _p(keep))
    return keep


def dcn_backward(x, offset, weight, grad_out, stride=1, pad=1, dil=1, dg=1, use_ref=False):
    """(grad_input, grad_offset, grad_weight) of the groups=1 deformable conv, through the column formulation:
    grad_col = W^T grad_out (fp64 GEMM), col2im / col2im_coord (oracle or the reference's kernels), grad_W = grad_out col^T."""
    x, offset, weight, grad_out = _f32(x), _f32(offset), _f32(weight), _f32(grad_out)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = grad_out.shape[2], grad_out.shape[3]
    go = grad_out.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64)           # [Cout, B*Ho*Wo]
    gcol = (weight.reshape(Cout, -1).astype(np.float64).T @ go).astype(np.float32)     # [C*taps, B*Ho*Wo]
    gcol = np.ascontiguousarray(gcol)
    gi = np.zeros_like(x); goff = np.zeros_like(offset)
    if use_ref:
        ref().ref_dcn_col2im(_p(gcol), _p(offset), B, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dg, _p(gi))
        ref().ref_dcn_col2im_coord(_p(gcol), _p(x), _p(offset), B, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil,
                                   dg, _p(goff))
    else:
        lib().orc_dcn_backward_input(_p(gcol), _p(x), _p(offset), B, C, H, W, kh, kw, pad, pad, stride, stride, dil,
                                     dil, dg, _p(gi), _p(goff))
    col = dcn_im2col(x, offset, kh, kw, pad, stride, dil, dg).reshape(C * kh * kw, -1).astype(np.float64)
    gw = (go @ col.T).reshape(weight.shape).astype(np.float32)
    return gi, goff, gw


def dcn_v2_im2col(x, offset, mask, kh, kw, pad, stride, dil, dg=1, use_ref=False):
    """Modulated columns [C*kh*kw, B, Ho, Wo] (deform_conv_cuda_kernel.cu:570-633).  use_ref: the reference's own kernel,
    driven per image with batch_size = 1 as deform_conv_cuda.cpp:540-545 does."""
    x, offset, mask = _f32(x), _f32(offset), _f32(mask)
    B, C, H, W = x.shape
    Ho, Wo = _odim(H, pad, dil, kh, stride), _odim(W, pad, dil, kw, stride)
    col = np.zeros((C * kh * kw, B, Ho, Wo), np.float32)
    if use_ref:
        for b in range(B):
            cb = np.zeros((C * kh * kw, Ho, Wo), np.float32)
            xb, ob, mb = (np.ascontiguousarray(a[b]) for a in (x, offset, mask))
            ref().ref_dcn_v2_im2col(_p(xb), _p(ob), _p(mb), C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dg, _p(cb))
            col[:, b] = cb
    else:
        lib().orc_dcn_v2_im2col(_p(x), _p(offset), _p(mask), B, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dg,
                                _p(col))
    return col


def dcn_v2_forward(x, offset, mask, weight, bias=None, stride=1, pad=1, dil=1, dg=1, use_ref=False):
    """ModulatedDeformConv forward (groups = 1) through the column formulation of deform_conv_cuda.cpp:490-567:
    out = W . col (fp64 GEMM) + bias, col from the oracle or (use_ref) the reference's modulated im2col kernel."""
    weight = _f32(weight)
    Cout, C, kh, kw = weight.shape
    col = dcn_v2_im2col(x, offset, mask, kh, kw, pad, stride, dil, dg, use_ref)
    _, B, Ho, Wo = col.shape
    out = weight.reshape(Cout, -1).astype(np.float64) @ col.reshape(C * kh * kw, -1).astype(np.float64)
    if bias is not None:
        out = out + _f32(bias).astype(np.float64)[:, None]
    return np.ascontiguousarray(out.reshape(Cout, B, Ho, Wo).transpose(1, 0, 2, 3)).astype(np.float32)


def dcn_v2_backward(x, offset, mask, weight, grad_out, stride=1, pad=1, dil=1, dg=1, use_ref=False):
    """(grad_input, grad_offset, grad_mask, grad_weight, grad_bias) of the groups = 1 ModulatedDeformConv through the
    column formulation of deform_conv_cuda.cpp:569-685: grad_col = W^T grad_out (fp64 GEMM), col2im_coord -> offset and
    mask gradients, col2im -> input gradient, grad_W = grad_out . col^T over the MODULATED columns, grad_bias = row sums.
    use_ref: the three column kernels are the reference's own (deform_conv_cuda_kernel.cu:570-767), per image."""
    x, offset, mask, weight, grad_out = _f32(x), _f32(offset), _f32(mask), _f32(weight), _f32(grad_out)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = grad_out.shape[2], grad_out.shape[3]
    go = grad_out.transpose(1, 0, 2, 3).reshape(Cout, -1).astype(np.float64)           # [Cout, B*Ho*Wo]
    gcol = (weight.reshape(Cout, -1).astype(np.float64).T @ go).astype(np.float32)     # [C*taps, B*Ho*Wo]
    gcol = np.ascontiguousarray(gcol).reshape(C * kh * kw, B, Ho, Wo)
    gi = np.zeros_like(x); goff = np.zeros_like(offset); gm = np.zeros_like(mask)
    if use_ref:
        for b in range(B):
            cb = np.ascontiguousarray(gcol[:, b])
            xb, ob, mb = (np.ascontiguousarray(a[b]) for a in (x, offset, mask))
            gib = np.zeros_like(xb); gob = np.zeros_like(ob); gmb = np.zeros_like(mb)
            ref().ref_dcn_v2_col2im_coord(_p(cb), _p(xb), _p(ob), _p(mb), C, H, W, kh, kw, pad, pad, stride, stride, dil,
                                          dil, dg, _p(gob), _p(gmb))
            ref().ref_dcn_v2_col2im(_p(cb), _p(ob), _p(mb), C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dg,
                                    _p(gib))
            gi[b], goff[b], gm[b] = gib, gob, gmb
    else:
        lib().orc_dcn_v2_backward_input(_p(gcol), _p(x), _p(offset), _p(mask), B, C, H, W, kh, kw, pad, pad, stride,
                                        stride, dil, dil, dg, _p(gi), _p(goff), _p(gm))
    col = dcn_v2_im2col(x, offset, mask, kh, kw, pad, stride, dil, dg, use_ref).reshape(C * kh * kw, -1).astype(np.float64)
    gw = (go @ col.T).reshape(weight.shape).astype(np.float32)
    gb = go.sum(axis=1).astype(np.float32)
    return gi, goff, gm, gw, gb


def box_iou_rotated(b1, b2, use_ref=False):
    a, b = _f32(b1), _f32(b2)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    fn = ref().ref_box_iou_rotated if use_ref else lib().orc_box_iou_rotated
    fn(_p(a), a.shape[0], _p(b), b.shape[0], _p(out))
    return out
