#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own functions (oracle/_ref/libref_orp.so, built by
oracle/build_ref.py from the sources under /root/reference).  Run in the build container only; the fixtures are
committed so that the GPU box (which has no /root/reference) can check against them.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import build_ref, orp_oracle as O  # noqa: E402
from orientedreppoints_amd import synthetic as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    assert build_ref.build(), "reference not available: cannot regenerate golden vectors"
    assert O.ref() is not None

    # ---- quad IoU (rnms devrIoU), plain and class-offset coordinates; NMS keep sets --------------------------
    g = {}
    d0 = S.gen_polys(160, 0).astype(np.float32)
    d1 = S.gen_polys(160, 1, clustered=True).astype(np.float32)
    d2, lab2 = S.gen_dense_scene(160, 2)
    d2 = d2.astype(np.float32)
    for name, d in (("uniform", d0), ("clustered", d1), ("offset", d2)):
        g["dets_" + name] = d
        g["iou_" + name] = O.ref_quad_iou_matrix(d, d)
    # keep sets (sorted-position space) for the three thresholds of SURVEY 8d, both IoU flavours
    for name, n, seed, kw in (("uniform", 500, 0, {}), ("clustered", 500, 10, dict(clustered=True))):
        d = S.gen_polys(n, seed, **kw).astype(np.float32)
        order = O.sort_order(d[:, 8])
        ds = np.ascontiguousarray(d[order])
        g["nms_dets_" + name] = d
        for thr in (0.1, 0.3, 0.4):
            g["nms_keep_%s_%02d_rnms" % (name, int(thr * 10))] = order[O.ref_nms_sorted(ds, thr, 0)]
            g["nms_keep_%s_%02d_poly" % (name, int(thr * 10))] = order[O.ref_nms_sorted(ds, thr, 1)]
    dd, lab = S.gen_dense_scene(600, 20)
    dd = dd.astype(np.float32)
    order = O.sort_order(dd[:, 8])
    g["nms_dets_dense"] = dd
    g["nms_keep_dense_04_rnms"] = order[O.ref_nms_sorted(np.ascontiguousarray(dd[order]), 0.4, 0)]
    np.savez_compressed(os.path.join(OUT, "quad_iou_nms.npz"), **g)

    # ---- fp64 polyiou ---------------------------------------------------------------------------------------
    d64 = S.gen_polys(60, 3, clustered=True)
    m = np.array([[O.ref_polyiou(d64[i, :8], d64[j, :8]) for j in range(60)] for i in range(60)])
    np.savez_compressed(os.path.join(OUT, "polyiou_f64.npz"), dets=d64, iou=m,
                        unit_p=np.array([0, 0, 1, 0, 1, 1, 0, 1.0]), unit_q=np.array([.5, .5, 1.5, .5, 1.5, 1.5, .5, 1.5]),
                        unit_iou=np.array(O.ref_polyiou([0, 0, 1, 0, 1, 1, 0, 1.0], [.5, .5, 1.5, .5, 1.5, 1.5, .5, 1.5])))

    # ---- poly_overlaps (5-param boxes) ----------------------------------------------------------------------
    rb = S.gen_rboxes(120, 5).astype(np.float32)
    rq = S.gen_rboxes(90, 6).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "poly_overlaps.npz"), boxes=rb, query=rq, iou=O.ref_poly_overlaps(rb, rq))

    # ---- box_iou_rotated (reference header, CPU branch) + the check value of SURVEY 8c ------------------------
    ba = S.gen_rboxes(100, 15).astype(np.float32)
    bb = S.gen_rboxes(80, 16).astype(np.float32)
    bb[:30, :2] = ba[:30, :2] + np.random.RandomState(17).normal(0, 10, (30, 2)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "box_iou_rotated.npz"), a=ba, b=bb, iou=O.box_iou_rotated(ba, bb, use_ref=True),
                        unit=O.box_iou_rotated(np.array([[0, 0, 10, 10, 0]], np.float32),
                                               np.array([[0.5, 0.5, 10, 10, 0]], np.float32), use_ref=True))

    # ---- minaerarect ----------------------------------------------------------------------------------------
    pts = S.gen_pointsets(1500, 7).astype(np.float32)
    grid = np.array([[x, y] for y in range(3) for x in range(0, 5, 2)], np.float32).reshape(1, 18)
    same = np.full((1, 18), 3.0, np.float32)
    line = np.array([[i, 2 * i] for i in range(9)], np.float32).reshape(1, 18)
    special = np.concatenate([grid, same, line], 0)
    np.savez_compressed(os.path.join(OUT, "minarearect.npz"), pts=pts, rect=O.ref_minarearect(pts), special=special,
                        special_rect=O.ref_minarearect(special))

    # ---- convex_iou / convex_giou ---------------------------------------------------------------------------
    gts = S.gen_gts(24, 8).astype(np.float32)
    ctr = np.repeat(gts.reshape(-1, 4, 2).mean(1), 20, axis=0)
    pts2 = S.gen_pointsets(480, 9, around=ctr).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "convex_iou.npz"), pts=pts2, gts=gts, iou=O.ref_convex_iou(pts2, gts))
    gts_al = np.repeat(gts, 20, axis=0)
    np.savez_compressed(os.path.join(OUT, "convex_giou.npz"), pts=pts2, gts=gts_al, out19=O.ref_convex_giou(pts2, gts_al))

    # ---- pointsJf: the 5 x 3 demo case of mmdet/ops/point_justify/test.py:10-17 + random ---------------------
    demo_p = np.array([[300., 300.], [400., 400.], [100., 100], [300, 250], [100, 0]], np.float32)
    demo_q = np.array([[200., 200., 400., 400., 500., 200., 400., 100.], [400., 400., 500., 500., 600., 300., 500., 200.],
                       [300., 300., 600., 700., 700., 700., 700., 100.]], np.float32)
    rng = np.random.RandomState(11)
    P = rng.uniform(0, 100, (400, 2)).astype(np.float32)
    Q = S.gen_polys(50, 12, wh=(8, 60))[:, :8].astype(np.float32) / 10.0
    np.savez_compressed(os.path.join(OUT, "points_justify.npz"), demo_p=demo_p, demo_q=demo_q,
                        demo_out=O.ref_points_justify(demo_p, demo_q), P=P, Q=Q, out=O.ref_points_justify(P, Q))

    # ---- chamfer + focal ------------------------------------------------------------------------------------
    a = rng.normal(0, 10, (64, 40, 2)).astype(np.float32)
    b = rng.normal(0, 10, (64, 40, 2)).astype(np.float32)
    d1_, i1_ = O.ref_chamfer_nn(a, b)
    d2_, i2_ = O.ref_chamfer_nn(b, a)
    x = rng.normal(0, 3, (800, 15)).astype(np.float32)
    t = rng.randint(0, 16, 800).astype(np.int64)
    gl = rng.uniform(0, 1, (800, 15)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "chamfer_focal.npz"), a=a, b=b, dist1=d1_, idx1=i1_, dist2=d2_, idx2=i2_,
                        logits=x, targets=t, d_losses=gl, focal_fwd=O.ref_focal_forward(x, t, 2.0, 0.25),
                        focal_bwd=O.ref_focal_backward(x, t, gl, 2.0, 0.25))
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
