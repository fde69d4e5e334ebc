#!/usr/bin/env python3
"""Golden vectors of the data-side box arithmetic, produced by executing the reference's own Python
where it lies under /root/reference (build container only):

  mmdet/core/bbox/transforms.py:273-295,401-500     rbbox_flip, rbox2poly, poly2rbox, get_best_begin_point
  mmdet/datasets/pipelines/transforms.py:85-270     RotateResize (scale draws, _resize_bboxes), RotateRandomFlip
  mmdet/datasets/pipelines/poly_transforms.py       PolyRandomFlip, PolyRandomRotate (box path, decisions, filter)
  mmdet/datasets/dota.py:32-82                      DotaDataset._parse_ann_info

cv2 / mmcv / matplotlib / pycocotools are not installed: the files are loaded one by one under stub modules.  Image
RESAMPLING inside those stubs is this repo's own imops (so image pixels are NOT a reference output and are not stored);
what is stored is what the reference's own numpy code computes: boxes, shapes, scale factors, random decisions.  cv2's
getRotationMatrix2D / transform are restated in the stub from OpenCV's documented formulas.  `np.float` (removed in
numpy 2) is aliased to float for the load.

    python tests/golden/make_golden_data.py        ->  tests/golden/data_py.npz
"""
import importlib.util
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

if not hasattr(np, 'float'):
    np.float = float

from orientedreppoints_amd.mmdet_datasets import imops  # noqa: E402


def stub(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        if '.' in name:
            parent, leaf = name.rsplit('.', 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, m)
    m.__dict__.update(attrs)
    return m


def load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    m.__package__ = modname.rsplit('.', 1)[0]
    sys.modules[modname] = m
    parent, leaf = modname.rsplit('.', 1)
    if parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    spec.loader.exec_module(m)
    return m


class _Reg(object):
    def __init__(self):
        self.module_dict = {}

    def register_module(self, cls=None):
        if cls is None:
            return self.register_module
        self.module_dict[cls.__name__] = cls
        return cls


def _cv2_transform(src, m):
    src = np.asarray(src, np.float64)
    m = np.asarray(m, np.float64)
    return src.dot(m[:, :2].T) + m[:, 2]


def setup():
    stub('mmcv', imrescale=imops.imrescale, imresize=imops.imresize, imflip=imops.imflip, impad=imops.impad,
         impad_to_multiple=imops.impad_to_multiple, imnormalize=imops.imnormalize,
         is_list_of=lambda seq, t: isinstance(seq, list) and all(isinstance(s, t) for s in seq),
         is_str=lambda s: isinstance(s, str))
    stub('mmcv.parallel', DataContainer=object)
    stub('cv2', INTER_LINEAR=1, INTER_NEAREST=0,
         getRotationMatrix2D=lambda center, angle, scale: imops.rotation_matrix_2d(center, angle, scale),
         transform=_cv2_transform,
         warpAffine=lambda img, m, dsize, flags=1: imops.warp_affine(img, m, dsize))
    stub('matplotlib')
    stub('matplotlib.pyplot', set_loglevel=lambda *a, **k: None)
    for name in ('albumentations', 'imagecorruptions', 'pycocotools', 'pycocotools.coco', 'pycocotools.mask'):
        stub(name)
    sys.modules['albumentations'].Compose = object
    sys.modules['imagecorruptions'].corrupt = None
    stub('mmdet')
    stub('mmdet.core')
    stub('mmdet.core.bbox')
    tr = load('mmdet.core.bbox.transforms', 'mmdet/core/bbox/transforms.py')
    sys.modules['mmdet.core'].poly2rbox = tr.poly2rbox
    sys.modules['mmdet.core'].rbox2poly = tr.rbox2poly
    stub('mmdet.core.evaluation')
    stub('mmdet.core.evaluation.bbox_overlaps', bbox_overlaps=None)
    stub('mmdet.datasets')
    stub('mmdet.datasets.registry', PIPELINES=_Reg(), DATASETS=_Reg())
    stub('mmdet.datasets.pipelines')
    pt = load('mmdet.datasets.pipelines.transforms', 'mmdet/datasets/pipelines/transforms.py')
    pp = load('mmdet.datasets.pipelines.poly_transforms', 'mmdet/datasets/pipelines/poly_transforms.py')
    stub('mmdet.datasets.coco', CocoDataset=object)
    dd = load('mmdet.datasets.dota', 'mmdet/datasets/dota.py')
    return tr, pt, pp, dd


def quads(n, seed, size=1024):
    """Random rotated rectangles as quads, float32 (+ a few general convex quads)."""
    rng = np.random.RandomState(seed)
    c = rng.uniform(60, size - 60, (n, 2))
    wh = rng.uniform(8, 120, (n, 2))
    a = rng.uniform(-np.pi, np.pi, n)
    base = np.array([[-.5, -.5], [.5, -.5], [.5, .5], [-.5, .5]])
    out = []
    for i in range(n):
        R = np.array([[np.cos(a[i]), -np.sin(a[i])], [np.sin(a[i]), np.cos(a[i])]])
        p = (base * wh[i]).dot(R.T) + c[i]
        if i % 5 == 4:
            p = p + rng.uniform(-3, 3, p.shape)
        out.append(np.roll(p, rng.randint(4), axis=0).reshape(-1))
    return np.array(out, dtype=np.float32)


def main():
    tr, pt, pp, dd = setup()
    out = {}

    # ---- transforms.py:401-500 -------------------------------------------------------------------------------------
    polys = quads(200, 1)
    polys[:4] = np.array([[10, 10, 50, 10, 50, 30, 10, 30], [10, 10, 10, 30, 50, 30, 50, 10],
                          [0, 0, 20, 0, 20, 20, 0, 20], [5, 5, 5, 5, 5, 5, 5, 5]], np.float32)     # axis-aligned, square, point
    rb = tr.poly2rbox(polys)
    out['geo_polys'], out['geo_poly2rbox'] = polys, rb
    out['geo_rbox2poly'] = tr.rbox2poly(rb)
    out['geo_best_begin'] = tr.get_best_begin_point(polys)
    for d in ('horizontal', 'vertical'):
        import torch
        out['geo_flip_' + d] = tr.rbbox_flip(torch.from_numpy(polys.reshape(25, 64)), (777, 1023, 3), d).numpy()
        out['geo_mapback_' + d] = tr.rbbox_mapping_back(torch.from_numpy(polys.reshape(25, 64)), (777, 1023, 3), 1.25, True, d).numpy()

    # ---- RotateResize: scale draws + box scaling/clamping ------------------------------------------------------------
    img = np.zeros((600, 800, 3), np.uint8)
    cases = [dict(img_scale=(1024, 1024), keep_ratio=True), dict(img_scale=(1333, 800), keep_ratio=True),
             dict(img_scale=[(1333, 768), (1333, 1280)], multiscale_mode='range', keep_ratio=True),
             dict(img_scale=[(1333, 768), (1333, 1280), (800, 800)], multiscale_mode='value', keep_ratio=True),
             dict(img_scale=(1000, 500), ratio_range=(0.5, 1.5), keep_ratio=True),
             # keep_ratio=False is not usable with 8-coordinate boxes in the reference ([n,8] * [4] broadcast error)
             dict(img_scale=(1024, 1024), keep_ratio=True, clamp_rbbox=False)]
    for i, kw in enumerate(cases):
        np.random.seed(100 + i)
        r = dict(img=img.copy(), img_shape=img.shape, gt_bboxes=quads(30, 10 + i, 820), bbox_fields=['gt_bboxes'])
        r = pt.RotateResize(**kw)(r)
        out['rr%d_scale' % i] = np.array(r['scale'])
        out['rr%d_shape' % i] = np.array(r['img_shape'])
        out['rr%d_factor' % i] = np.asarray(r['scale_factor'], np.float64)
        out['rr%d_boxes' % i] = r['gt_bboxes']
    np.random.seed(7)
    r = dict(img=img.copy(), img_shape=img.shape, gt_bboxes=quads(30, 3, 820), bbox_fields=['gt_bboxes'], scale=(1024, 1024))
    r = pp.PolyResize(img_scale=[(1333, 768), (1333, 1280)], multiscale_mode='range')(r)
    out['pr_scale'], out['pr_shape'], out['pr_boxes'] = np.array(r['scale']), np.array(r['img_shape']), r['gt_bboxes']

    # ---- flips ---------------------------------------------------------------------------------------------------------------
    np.random.seed(11)
    dec, boxes = [], []
    for k in range(12):
        r = dict(img=img.copy(), img_shape=img.shape, gt_bboxes=quads(8, 50 + k, 600), bbox_fields=['gt_bboxes'])
        r = pt.RotateRandomFlip(flip_ratio=0.5, direction=['horizontal', 'vertical'])(r)
        dec.append([int(r['flip']), int(str(np.asarray(r['flip_direction']).reshape(-1)[0]) == 'vertical')])
        boxes.append(r['gt_bboxes'])
    out['rf_dec'], out['rf_boxes'] = np.array(dec), np.array(boxes)
    np.random.seed(12)
    random.seed(12)
    dec, boxes = [], []
    for k in range(12):
        r = dict(img=img.copy(), img_shape=img.shape, gt_bboxes=quads(8, 80 + k, 600), bbox_fields=['gt_bboxes'])
        r = pp.PolyRandomFlip(flip_ratio=0.5)(r)
        dec.append([int(r['flip']), int(r['flip_direction'] == 'vertical')])
        boxes.append(r['gt_bboxes'])
    out['pf_dec'], out['pf_boxes'] = np.array(dec), np.array(boxes)

    # ---- PolyRandomRotate: decisions, matrices, surviving boxes ----------------------------------------------------------
    for ab in (0,):       # auto_bound=True raises in the reference (tuple indexed with [None, None, :], :398-399)
        np.random.seed(21 + ab)
        random.seed(21 + ab)
        t = pp.PolyRandomRotate(rotate_ratio=0.7, angles_range=180, auto_bound=bool(ab))
        for k in range(8):
            r = dict(img=img.copy(), img_shape=img.shape, gt_bboxes=quads(12, 120 + k, 600),
                     gt_labels=np.arange(12, dtype=np.int64) % 15 + 1, bbox_fields=['gt_bboxes'])
            r = t(r)
            tag = 'rot%d_%d_' % (ab, k)
            out[tag + 'none'] = np.array(int(r is None))
            out[tag + 'rm'] = np.asarray(t.rm_coords, np.float64)
            out[tag + 'rm_img'] = np.asarray(t.rm_image, np.float64)
            if r is not None:
                out[tag + 'angle'] = np.array(r['rotate_angle'], np.float64)
                out[tag + 'shape'] = np.array(r['img_shape'])
                out[tag + 'boxes'], out[tag + 'labels'] = r['gt_bboxes'], r['gt_labels']

    # ---- DotaDataset._parse_ann_info ----------------------------------------------------------------------------------------
    ds = dd.DotaDataset.__new__(dd.DotaDataset)
    ds.cat2label = {cid: i + 1 for i, cid in enumerate(range(1, 16))}
    q = quads(9, 200)
    anns = [dict(bbox=[float(v) for v in q[i]], category_id=int(i % 15 + 1), iscrowd=int(i == 3), ignore=(i == 5),
                 segmentation=[[float(v) for v in q[i]]], area=10.0) for i in range(9)]
    a = ds._parse_ann_info(dict(filename='P0001.jpg'), anns)
    out['dota_in'] = q
    out['dota_bboxes'], out['dota_labels'], out['dota_ignore'] = a['bboxes'], a['labels'], a['bboxes_ignore']
    a0 = ds._parse_ann_info(dict(filename='P0002.jpg'), [])
    out['dota_empty_bboxes'], out['dota_empty_labels'] = a0['bboxes'], a0['labels']
    assert a['seg_map'] == 'P0001.png'

    path = os.path.join(HERE, 'data_py.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
