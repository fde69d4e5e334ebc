"""convex_iou timing with the assigner's real layout: all FPN levels' point sets in row-major order (dev aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.mmdet_ops import convex_iou
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ar = []
for s in (8, 16, 32, 64, 128):
    n = 1024 // s
    yy, xx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    ar.append(np.stack([xx.reshape(-1) * s + s / 2.0, yy.reshape(-1) * s + s / 2.0], 1))
around = np.concatenate(ar)
pts = torch.from_numpy(np.ascontiguousarray(S.gen_pointsets(len(around), 6, around=around), np.float32)).to(dev)
for k in (8, 64, 256):
    gts = torch.from_numpy(S.gen_gts(k, 3).astype(np.float32)).to(dev)
    us = timeit(lambda: convex_iou(pts, gts))
    print("convex_iou grid-ordered %d x %d: %.1f us (%.2f ns/pair)" % (pts.size(0), k, us, us * 1e3 / (pts.size(0) * k)))
# convex_giou (aligned pairs, value + 18 gradients): the loss-side shapes
from orientedreppoints_amd.mmdet_ops import convex_giou
for P in (500, 5000, 20000):
    pp = torch.from_numpy(S.gen_pointsets(P, 2).astype(np.float32)).to(dev)
    gg = torch.from_numpy(S.gen_gts(P, 3).astype(np.float32)).to(dev)
    us = timeit(lambda: convex_giou(pp, gg))
    print("convex_giou %d pairs: %.1f us (%.1f ns/pair)" % (P, us, us * 1e3 / P))
