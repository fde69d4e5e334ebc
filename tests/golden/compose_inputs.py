"""Seeded inputs of the composition goldens (numpy only, legacy RandomState => identical on every machine).

Imported by tests/golden/make_golden_compose.py (which feeds them to the reference's own Python) and by the GPU tests
(which feed the same arrays to the HIP path).  Nothing here is part of the product.
"""
import numpy as np

STRIDES = (8, 16, 32, 64, 128)
NUM_POINTS = 9
NUM_CLS = 15


def level_sizes(img_size):
    return [img_size // s for s in STRIDES]


def _grid_offsets(rng, f, spread):
    """[18, f, f] (y,x)-interleaved 3x3 pattern in grid units: a rotated, scaled, jittered 3x3 grid per location."""
    g = np.array([[y, x] for y in (-1.0, 0.0, 1.0) for x in (-1.0, 0.0, 1.0)])          # 9 x (y, x)
    sy = rng.uniform(spread[0], spread[1], size=(f, f, 1))
    sx = rng.uniform(spread[0], spread[1], size=(f, f, 1))
    th = rng.uniform(-np.pi / 2, np.pi / 2, size=(f, f, 1))
    py = g[None, None, :, 0] * sy
    px = g[None, None, :, 1] * sx
    c, s = np.cos(th), np.sin(th)
    y = s * px + c * py + rng.normal(0, 0.15, size=(f, f, 9))
    x = c * px - s * py + rng.normal(0, 0.15, size=(f, f, 9))
    yx = np.stack([y, x], axis=-1).reshape(f, f, 18)
    return np.ascontiguousarray(yx.transpose(2, 0, 1)).astype(np.float32)


def postprocess_scene(img_size, seed, logit_mean=-8.0, logit_std=1.0, spread=(0.4, 2.5), obj_density=1.0 / 40, peak=(6.0, 9.0)):
    """One image of test-time head outputs: cls_scores[l] [15,H,W] logits, pts_refine[l] [18,H,W] offsets.

    Background logits ~ N(mean, std); on top, "objects" (a centre, a class, a size per level): the locations around an
    object raise that class's logit (Gaussian bump) and regress their nine points onto the same object, so that -- as
    with a trained detector -- clusters of same-class, heavily overlapping boxes reach the NMS."""
    rng = np.random.RandomState(seed)
    cls, pts = [], []
    for f in level_sizes(img_size):
        c = rng.normal(logit_mean, logit_std, size=(NUM_CLS, f, f))
        p = _grid_offsets(rng, f, spread).astype(np.float64)
        n_obj = int(round(f * f * obj_density)) if obj_density > 0 else 0
        yy, xx = np.meshgrid(np.arange(f), np.arange(f), indexing='ij')
        for _ in range(n_obj):
            oy, ox = rng.uniform(0, f, size=2)
            k = rng.randint(0, NUM_CLS)
            rad = rng.uniform(1.0, 2.5)
            amp = rng.uniform(peak[0], peak[1])
            d2 = (yy - oy) ** 2 + (xx - ox) ** 2
            c[k] += amp * np.exp(-d2 / (2 * rad * rad))
            near = d2 <= (2 * rad) ** 2
            if near.any():
                sy, sx = rng.uniform(spread[0], spread[1], size=2)
                th = rng.uniform(-np.pi / 2, np.pi / 2)
                g = np.array([[y, x] for y in (-1.0, 0.0, 1.0) for x in (-1.0, 0.0, 1.0)])
                py, px = g[:, 0] * sy, g[:, 1] * sx
                oyx = np.stack([np.sin(th) * px + np.cos(th) * py, np.cos(th) * px - np.sin(th) * py], 1)   # 9 x (y, x)
                ys, xs = np.nonzero(near)
                for y, x in zip(ys, xs):
                    tgt = oyx + np.array([oy - y, ox - x]) + rng.normal(0, 0.2, size=(9, 2))
                    p[:, y, x] = tgt.reshape(18)
        cls.append(c.astype(np.float32))
        pts.append(np.ascontiguousarray(p).astype(np.float32))
    return cls, pts


def gt_quads(k, seed, img_size, wh):
    """[k,8] float32 gt quads inside the image + labels 1..15."""
    rng = np.random.RandomState(seed)
    cx = rng.uniform(0.1 * img_size, 0.9 * img_size, size=k)
    cy = rng.uniform(0.1 * img_size, 0.9 * img_size, size=k)
    w = rng.uniform(wh[0], wh[1], size=k)
    h = rng.uniform(wh[0], wh[1], size=k)
    th = rng.uniform(-np.pi / 2, np.pi / 2, size=k)
    c, s = np.cos(th), np.sin(th)
    dx = np.stack([-w / 2, w / 2, w / 2, -w / 2], axis=1)
    dy = np.stack([-h / 2, -h / 2, h / 2, h / 2], axis=1)
    x = cx[:, None] + c[:, None] * dx - s[:, None] * dy
    y = cy[:, None] + s[:, None] * dx + c[:, None] * dy
    quads = np.stack([x, y], axis=2).reshape(-1, 8).astype(np.float32)
    labels = rng.randint(1, NUM_CLS + 1, size=k).astype(np.int64)
    return quads, labels


def loss_case(img_size, num_gts, seed, channels=256, wh=(12.0, 90.0)):
    """Head outputs + targets of one training batch (B = len(num_gts) images):
    cls_scores[l] [B,15,H,W], pts_init[l] [B,18,H,W], pts_refine[l] [B,18,H,W], feats[l] [B,C,H,W],
    gts[i] [K_i,8], labels[i] [K_i]."""
    rng = np.random.RandomState(seed)
    B = len(num_gts)
    cls, init, refine, feats = [], [], [], []
    for f in level_sizes(img_size):
        cls.append(rng.normal(-2.0, 1.5, size=(B, NUM_CLS, f, f)).astype(np.float32))
        pi = np.stack([_grid_offsets(rng, f, (0.5, 3.0)) for _ in range(B)], 0)
        init.append(pi)
        refine.append((pi + rng.normal(0, 0.3, size=pi.shape)).astype(np.float32))
        feats.append(rng.normal(0, 1.0, size=(B, channels, f, f)).astype(np.float32))
    gts, labels = [], []
    for i, k in enumerate(num_gts):
        q, l = gt_quads(k, seed * 131 + i, img_size, wh)
        gts.append(q)
        labels.append(l)
    return dict(cls=cls, init=init, refine=refine, feats=feats, gts=gts, labels=labels)


def img_meta(img_size):
    return dict(img_shape=(img_size, img_size, 3), pad_shape=(img_size, img_size, 3), scale_factor=1.0, flip=False)
