"""Timing of the remaining hot-path ops at configs[2] shapes (dev aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.mmdet_ops import (box_iou_rotated, ChamferDistance2D, points_in_quad_aligned, sigmoid_focal_loss,
                                             convex_giou, minaerarect)
from orientedreppoints_amd.mmdet_ops.apaa import point_assign, max_iou_assign, apaa_select, apaa_feature_dissimilarity
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
rb = t(S.gen_rboxes(2000, 1))
print("box_iou_rotated 2000x2000: %.1f us" % timeit(lambda: box_iou_rotated(rb, rb)))
P = 5000
pts = t(S.gen_pointsets(P, 2)); gts = t(S.gen_gts(P, 3))
print("convex_giou P=%d: %.1f us" % (P, timeit(lambda: convex_giou(pts, gts))))
print("minaerarect P=%d: %.1f us" % (P, timeit(lambda: minaerarect(pts))))
a = torch.rand(P, 40, 2, device=dev) * 100; b = torch.rand(P, 40, 2, device=dev) * 100
print("chamfer P=%d 40x40: %.1f us" % (P, timeit(lambda: ChamferDistance2D(a, b))))
print("points_in_quad_aligned P=%d: %.1f us" % (P, timeit(lambda: points_in_quad_aligned(pts, gts))))
N = 2 * 21824
logits = torch.randn(N, 15, device=dev); labels = torch.randint(0, 16, (N,), device=dev)
print("sigmoid_focal_loss fwd N=%d: %.1f us" % (N, timeit(lambda: sigmoid_focal_loss(logits, labels, 2.0, 0.25))))
ar = []
for s in (8, 16, 32, 64, 128):
    n = 1024 // s
    yy, xx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ar.append(np.stack([xx.reshape(-1) * s, yy.reshape(-1) * s, np.full(n * n, s)], 1))
points = t(np.concatenate(ar))
g64 = t(S.gen_gts(64, 5))
print("point_assign N=21824 K=64: %.1f us" % timeit(lambda: point_assign(points, g64)))
ov = torch.rand(21824, 64, device=dev) * 0.3              # [N, K] point-major, what convex_iou produces
print("max_iou_assign N=21824 K=64: %.1f us" % timeit(lambda: max_iou_assign(ov, 0.1, 0.1)))
q = torch.rand(P, device=dev); pg = torch.randint(1, 65, (P,), device=dev); pl = torch.randint(0, 5, (P,), device=dev, dtype=torch.int32)
print("apaa_select P=%d K=64: %.1f us" % (P, timeit(lambda: apaa_select(q, pg, pl, 64, 5))))
feats = [torch.randn(2, 256, 1024 // s, 1024 // s, device=dev) for s in (8, 16, 32, 64, 128)]
ii = torch.randint(0, 2, (P,), device=dev, dtype=torch.int32)
print("apaa_feature_dissimilarity P=%d: %.1f us" % (P, timeit(lambda: apaa_feature_dissimilarity(feats, [8, 16, 32, 64, 128], pts, ii, pl))))
