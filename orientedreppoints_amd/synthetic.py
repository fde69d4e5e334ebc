"""Seeded synthetic inputs for the hot path.  numpy only; used by tests/ and bench.py.

Shapes follow the DOTA configs (configs/dota/orientedrepoints_r50_demo.py): 1024x1024 patches, strides
(8,16,32,64,128), 9 points per location, 15 foreground classes.
"""
import numpy as np

IMG = 1024.0


def _corners(cx, cy, w, h, th):
    """corners = centre + R(theta) * (+-w/2, +-h/2) in order (-,-),(+,-),(+,+),(-,+)."""
    c, s = np.cos(th), np.sin(th)
    dx = np.stack([-w / 2, w / 2, w / 2, -w / 2], axis=1)
    dy = np.stack([-h / 2, -h / 2, h / 2, h / 2], axis=1)
    x = cx[:, None] + c[:, None] * dx - s[:, None] * dy
    y = cy[:, None] + s[:, None] * dx + c[:, None] * dy
    return np.stack([x, y], axis=2).reshape(-1, 8)


def gen_polys(n, seed, clustered=False, wh=(8.0, 128.0)):
    """dets[n,9] float64 = 8 corner coords + score   (SURVEY 8d `gen(n, seed)`).

    clustered=True draws the centres around 20 hubs (sigma = 24 px) so that suppression actually happens.
    """
    rng = np.random.RandomState(seed)
    if clustered:
        hubs = rng.uniform(0, IMG, size=(20, 2))
        which = rng.randint(0, 20, size=n)
        ctr = hubs[which] + rng.normal(0, 24.0, size=(n, 2))
        cx, cy = ctr[:, 0], ctr[:, 1]
    else:
        cx = rng.uniform(0, IMG, size=n)
        cy = rng.uniform(0, IMG, size=n)
    w = rng.uniform(wh[0], wh[1], size=n)
    h = rng.uniform(wh[0], wh[1], size=n)
    th = rng.uniform(-np.pi / 2, np.pi / 2, size=n)
    score = rng.uniform(0.05, 1.0, size=n)
    return np.concatenate([_corners(cx, cy, w, h, th), score[:, None]], axis=1)


def gen_rboxes(n, seed, wh=(8.0, 128.0)):
    """(cx, cy, w, h, theta[rad]) float64 boxes for poly_overlaps / box_iou_rotated."""
    rng = np.random.RandomState(seed)
    cx = rng.uniform(0, IMG, size=n)
    cy = rng.uniform(0, IMG, size=n)
    w = rng.uniform(wh[0], wh[1], size=n)
    h = rng.uniform(wh[0], wh[1], size=n)
    th = rng.uniform(-np.pi / 2, np.pi / 2, size=n)
    return np.stack([cx, cy, w, h, th], axis=1)


def class_offset(dets, labels):
    """The class-offset trick of multiclass_rnms (mmdet/core/post_processing/bbox_nms.py:156-158):
    coords + label * (max_coordinate + 1)."""
    d = np.array(dets, copy=True)
    mx = d[:, :8].max()
    d[:, :8] = d[:, :8] + (labels[:, None].astype(d.dtype) * (mx + 1))
    return d


def gen_dense_scene(n, seed, num_classes=15, clustered=True):
    """config-4 style stress: n dets with random labels folded in by the class-offset trick."""
    d = gen_polys(n, seed, clustered=clustered)
    rng = np.random.RandomState(seed + 1000)
    labels = rng.randint(0, num_classes, size=n)
    return class_offset(d, labels), labels


def gen_pointsets(n, seed, around=None):
    """[n,18] (x,y)-interleaved 9-point sets: regular 3x3 grid x U(1,6)*scale + N(0,0.5) jitter + random rotation
    (SURVEY 8d 'point sets').  If `around` ([n,2] centres) is given the sets are centred there."""
    rng = np.random.RandomState(seed)
    g = np.array([[x, y] for y in (-1.0, 0.0, 1.0) for x in (-1.0, 0.0, 1.0)])  # 9x2
    sx = rng.uniform(1, 6, size=(n, 1)) * rng.uniform(1, 8, size=(n, 1))
    sy = rng.uniform(1, 6, size=(n, 1)) * rng.uniform(1, 8, size=(n, 1))
    th = rng.uniform(-np.pi / 2, np.pi / 2, size=(n, 1))
    px = g[None, :, 0] * sx + rng.normal(0, 0.5, size=(n, 9))
    py = g[None, :, 1] * sy + rng.normal(0, 0.5, size=(n, 9))
    c, s = np.cos(th), np.sin(th)
    x = c * px - s * py
    y = s * px + c * py
    if around is None:
        around = rng.uniform(0, IMG, size=(n, 2))
    x = x + around[:, :1]
    y = y + around[:, 1:2]
    return np.stack([x, y], axis=2).reshape(n, 18)


def gen_gts(k, seed):
    """[k,8] gt quads (same generator as the dets, without the score)."""
    return gen_polys(k, seed + 77)[:, :8]
