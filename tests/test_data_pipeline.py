"""CPU: data-side rotated-box operations and DotaDataset against tests/golden/data_py.npz, which
tests/golden/make_golden_data.py produced by executing the reference's own Python (transforms.py, pipelines/transforms.py,
pipelines/poly_transforms.py, datasets/dota.py).  Box arithmetic, shapes, scale factors and seeded random decisions are
compared exactly; min_area_rect / box_points (cv2 stand-ins) and image resampling are checked by their properties."""
import json
import os
import random

import numpy as np
import pytest
import torch

from orientedreppoints_amd.mmdet_datasets import dota, geometry as G, imops, pipelines as P

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "data_py.npz"))
IMG = np.zeros((600, 800, 3), np.uint8)


def quads(n, seed, size=1024):                       # same generator as make_golden_data.quads
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


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a, b, equal_nan=True)


def test_poly2rbox_rbox2poly_best_begin_point_exact():
    polys = GOLD['geo_polys']
    rb = G.poly2rbox(polys)
    assert same(rb, GOLD['geo_poly2rbox'])
    assert same(G.rbox2poly(rb), GOLD['geo_rbox2poly'])
    assert same(G.get_best_begin_point(polys), GOLD['geo_best_begin'])
    assert np.all(rb[:, 2] >= rb[:, 3])
    assert np.all((rb[:, 4] >= -np.pi / 4) & (rb[:, 4] < 3 * np.pi / 4))


def test_rbbox_flip_and_mapping_back_exact():
    polys = GOLD['geo_polys'].reshape(25, 64)
    for d in ('horizontal', 'vertical'):
        assert same(P.rbbox_flip(polys, (777, 1023, 3), d), GOLD['geo_flip_' + d])
        assert np.allclose(P.rbbox_flip(P.rbbox_flip(polys, (777, 1023, 3), d), (777, 1023, 3), d), polys, atol=1e-3)
        back = P.rbbox_mapping_back(torch.from_numpy(polys), (777, 1023, 3), 1.25, True, d)
        assert same(back.numpy(), GOLD['geo_mapback_' + d])
    with pytest.raises(ValueError):
        P.rbbox_flip(polys, (10, 10, 3), 'diagonal')


RR_CASES = [dict(img_scale=(1024, 1024), keep_ratio=True), dict(img_scale=(1333, 800), keep_ratio=True),
            dict(img_scale=[(1333, 768), (1333, 1280)], multiscale_mode='range', keep_ratio=True),
            dict(img_scale=[(1333, 768), (1333, 1280), (800, 800)], multiscale_mode='value', keep_ratio=True),
            dict(img_scale=(1000, 500), ratio_range=(0.5, 1.5), keep_ratio=True),
            dict(img_scale=(1024, 1024), keep_ratio=True, clamp_rbbox=False)]


@pytest.mark.parametrize("i", range(len(RR_CASES)))
def test_rotate_resize_scale_draws_and_boxes(i):
    np.random.seed(100 + i)
    r = dict(img=IMG.copy(), img_shape=IMG.shape, gt_bboxes=quads(30, 10 + i, 820), bbox_fields=['gt_bboxes'])
    r = P.RotateResize(**RR_CASES[i])(r)
    assert tuple(r['scale']) == tuple(GOLD['rr%d_scale' % i])
    assert tuple(r['img_shape']) == tuple(GOLD['rr%d_shape' % i]) == r['img'].shape
    assert np.array_equal(np.asarray(r['scale_factor'], np.float64), GOLD['rr%d_factor' % i])
    assert same(r['gt_bboxes'], GOLD['rr%d_boxes' % i])


def test_poly_resize_redraws_a_preset_scale():
    np.random.seed(7)
    r = dict(img=IMG.copy(), img_shape=IMG.shape, gt_bboxes=quads(30, 3, 820), bbox_fields=['gt_bboxes'], scale=(1024, 1024))
    r = P.PolyResize(img_scale=[(1333, 768), (1333, 1280)], multiscale_mode='range')(r)
    assert tuple(r['scale']) == tuple(GOLD['pr_scale']) and tuple(r['img_shape']) == tuple(GOLD['pr_shape'])
    assert same(r['gt_bboxes'], GOLD['pr_boxes'])


def test_random_flips_make_the_reference_decisions():
    np.random.seed(11)
    for k in range(12):
        r = dict(img=IMG.copy(), img_shape=IMG.shape, gt_bboxes=quads(8, 50 + k, 600), bbox_fields=['gt_bboxes'])
        r = P.RotateRandomFlip(flip_ratio=0.5, direction=['horizontal', 'vertical'])(r)
        d = str(np.asarray(r['flip_direction']).reshape(-1)[0])
        assert [int(r['flip']), int(d == 'vertical')] == GOLD['rf_dec'][k].tolist()
        assert same(r['gt_bboxes'], GOLD['rf_boxes'][k])
    np.random.seed(12)
    random.seed(12)
    for k in range(12):
        r = dict(img=IMG.copy(), img_shape=IMG.shape, gt_bboxes=quads(8, 80 + k, 600), bbox_fields=['gt_bboxes'])
        r = P.PolyRandomFlip(flip_ratio=0.5)(r)
        assert [int(r['flip']), int(r['flip_direction'] == 'vertical')] == GOLD['pf_dec'][k].tolist()
        assert same(r['gt_bboxes'], GOLD['pf_boxes'][k])
    assert GOLD['rf_dec'][:, 0].sum() not in (0, 12) and GOLD['pf_dec'][:, 0].sum() not in (0, 12)


def test_poly_random_rotate_matrices_and_surviving_boxes():
    np.random.seed(21)
    random.seed(21)
    t = P.PolyRandomRotate(rotate_ratio=0.7, angles_range=180, auto_bound=False)
    rotated = 0
    for k in range(8):
        r = dict(img=IMG.copy(), img_shape=IMG.shape, gt_bboxes=quads(12, 120 + k, 600),
                 gt_labels=np.arange(12, dtype=np.int64) % 15 + 1, bbox_fields=['gt_bboxes'])
        r = t(r)
        tag = 'rot0_%d_' % k
        assert int(r is None) == int(GOLD[tag + 'none'])
        assert np.array_equal(t.rm_coords, GOLD[tag + 'rm']) and np.array_equal(t.rm_image, GOLD[tag + 'rm_img'])
        if r is not None:
            assert float(r['rotate_angle']) == float(GOLD[tag + 'angle'])
            assert tuple(r['img_shape']) == tuple(GOLD[tag + 'shape'])
            assert same(r['gt_bboxes'], GOLD[tag + 'boxes']) and same(r['gt_labels'], GOLD[tag + 'labels'])
            rotated += int(r['rotate'])
            assert len(r['gt_bboxes']) <= 12
    assert rotated >= 2


def test_dota_parse_ann_info_exact():
    ds = dota.DotaDataset.__new__(dota.DotaDataset)
    ds.cat2label = {cid: i + 1 for i, cid in enumerate(range(1, 16))}
    q = GOLD['dota_in']
    anns = [dict(bbox=[float(v) for v in q[i]], category_id=int(i % 15 + 1), iscrowd=int(i == 3), ignore=(i == 5),
                 segmentation=[[float(v) for v in q[i]]], area=10.0) for i in range(9)]
    a = ds._parse_ann_info(dict(filename='P0001.jpg'), anns)
    assert same(a['bboxes'], GOLD['dota_bboxes']) and same(a['labels'], GOLD['dota_labels'])
    assert same(a['bboxes_ignore'], GOLD['dota_ignore']) and a['seg_map'] == 'P0001.png'
    a0 = ds._parse_ann_info(dict(filename='P0002.jpg'), [])
    assert same(a0['bboxes'], GOLD['dota_empty_bboxes']) and same(a0['labels'], GOLD['dota_empty_labels'])


# ---- cv2 stand-ins: properties ---------------------------------------------------------------------------------------
def test_min_area_rect_is_a_tight_minimum_rectangle():
    rng = np.random.RandomState(5)
    for k in range(200):
        pts = rng.randint(0, 400, (4 if k % 2 else 9, 2)).astype(np.float64)
        (cx, cy), (w, h), ang = G.min_area_rect(pts)
        assert 0.0 < ang <= 90.0
        box = G.box_points(((cx, cy), (w, h), ang)).astype(np.float64)
        a = np.deg2rad(ang)
        u, v = np.array([np.cos(a), np.sin(a)]), np.array([-np.sin(a), np.cos(a)])
        pu, pv = (pts - [cx, cy]).dot(u), (pts - [cx, cy]).dot(v)
        assert np.all(np.abs(pu) <= w / 2 + 1e-6) and np.all(np.abs(pv) <= h / 2 + 1e-6)      # encloses
        assert np.isclose(np.abs(pu).max(), w / 2, atol=1e-6) and np.isclose(np.abs(pv).max(), h / 2, atol=1e-6)   # tight
        assert np.allclose(box.mean(0), [cx, cy], atol=1e-3)
        # minimal over a fine sweep of orientations
        best = min((np.ptp(pts.dot([np.cos(t), np.sin(t)])) * np.ptp(pts.dot([-np.sin(t), np.cos(t)]))
                    for t in np.linspace(0, np.pi / 2, 721)))
        assert w * h <= best + 1e-6 * max(1.0, best)


def test_min_area_rect_of_a_rotated_rectangle_returns_its_corners():
    rb = np.array([[300.0, 200.0, 120.0, 40.0, 0.3], [100.0, 100.0, 50.0, 50.0, -0.2], [64.0, 64.0, 20.0, 90.0, 1.2]])
    for r in rb:
        c, s = np.cos(r[4]), np.sin(r[4])
        base = np.array([[-.5, -.5], [.5, -.5], [.5, .5], [-.5, .5]]) * r[2:4]
        pts = base.dot(np.array([[c, s], [-s, c]])) + r[:2]
        box = G.box_points(G.min_area_rect(pts)).astype(np.float64)
        d = np.linalg.norm(box[:, None] - pts[None], axis=-1)
        assert np.all(d.min(1) < 1e-3) and sorted(d.argmin(1)) == [0, 1, 2, 3]
    # axis-aligned: OpenCV >= 4.5.1 reports 90 degrees
    (_, _), (w, h), ang = G.min_area_rect(np.array([[0, 0], [40, 0], [40, 10], [0, 10]]))
    assert ang == 90.0 and (w, h) == (10.0, 40.0)


def test_correct_box_keeps_the_first_corner_first():
    q = quads(40, 9, 800)
    r = P.CorrectBox(correct_rbbox=True, refine_rbbox=True)(dict(gt_bboxes=q.copy()))
    out = r['gt_bboxes']
    assert out.dtype == np.float32 and out.shape == (40, 8)
    for a, b in zip(q, out):
        ai = a.astype(np.int64).reshape(4, 2)
        bb = b.reshape(4, 2)
        d = np.linalg.norm(bb - ai[0], axis=1)
        assert d[0] == d.min()
        e = np.linalg.norm(bb - np.roll(bb, -1, 0), axis=1)                       # a rectangle: equal opposite edges, right angles
        assert abs(e[0] - e[2]) < 1e-2 and abs(e[1] - e[3]) < 1e-2
        assert abs(np.dot(bb[1] - bb[0], bb[2] - bb[1])) < 1e-1 * max(1.0, e[0] * e[1])
    assert P.PIPELINES.get('CorrectRBBox') is not None


# ---- image operations (not reference outputs: cv2 is absent everywhere we can run) ------------------------------------------
def test_image_ops_properties():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (60, 80, 3)).astype(np.uint8)
    assert np.array_equal(imops.imresize(img, (80, 60)), img)
    out, sf = imops.imrescale(img, (1024, 1024), return_scale=True)
    assert out.shape == (768, 1024, 3) and sf == 12.8
    const = np.full((33, 47, 3), 77, np.uint8)
    assert np.all(imops.imresize(const, (100, 71)) == 77)
    assert np.array_equal(imops.imflip(imops.imflip(img, 'horizontal'), 'horizontal'), img)
    assert np.array_equal(imops.imflip(img, 'vertical'), img[::-1])
    p = imops.impad_to_multiple(img, 32)
    assert p.shape == (64, 96, 3) and np.array_equal(p[:60, :80], img) and p[60:].sum() == 0
    n = imops.imnormalize(img, [1, 2, 3], [2, 2, 2], to_rgb=True)
    assert n.dtype == np.float32 and np.allclose(n[..., 0], (img[..., 2].astype(np.float32) - 1) / 2)
    # rotation by 90 degrees about the centre of a square image with the -0.5 pixel-centre offset == np.rot90
    sq = rng.randint(0, 256, (64, 64, 3)).astype(np.uint8)
    m = imops.rotation_matrix_2d((32 - 0.5, 32 - 0.5), 90, 1)
    assert np.array_equal(imops.warp_affine(sq, m, (64, 64)), np.rot90(sq))
    assert np.array_equal(imops.warp_affine(sq, imops.rotation_matrix_2d((10, 10), 0, 1), (64, 64)), sq)


def _write_dataset(tmp_path, n_img=3):
    rng = np.random.RandomState(3)
    images, anns = [], []
    aid = 0
    for i in range(n_img):
        h, w = (96, 128) if i != 1 else (128, 96)
        name = 'P%04d.npy' % i
        np.save(os.path.join(tmp_path, name), rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
        images.append(dict(id=10 + i, file_name=name, width=w, height=h))
        if i == 2:
            continue                                          # image without annotations: filtered in train mode
        for q in quads(5, 40 + i, 96):
            anns.append(dict(id=aid, image_id=10 + i, category_id=int(aid % 15 + 1), bbox=[float(v) for v in q],
                             iscrowd=0, area=1.0, segmentation=[]))
            aid += 1
    cats = [dict(id=i + 1, name=n) for i, n in enumerate(dota.DotaDataset.CLASSES)]
    ann_file = os.path.join(tmp_path, 'ann.json')
    with open(ann_file, 'w') as f:
        json.dump(dict(images=images, annotations=anns, categories=cats), f)
    return ann_file


IMG_NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def test_dota_dataset_train_and_test_pipelines_end_to_end(tmp_path):
    tmp_path = str(tmp_path)
    ann_file = _write_dataset(tmp_path)
    train_pipeline = [                                          # configs/dota/*.py train_pipeline
        dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
        dict(type='CorrectBox', correct_rbbox=True, refine_rbbox=True),
        dict(type='RotateResize', img_scale=[(200, 128), (200, 192)], multiscale_mode='range', keep_ratio=True, clamp_rbbox=False),
        dict(type='RotateRandomFlip', flip_ratio=0.5), dict(type='Normalize', **IMG_NORM), dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels'])]
    ds = dota.build_dataset(dict(type='DotaDataset', ann_file=ann_file, img_prefix=tmp_path, pipeline=train_pipeline))
    assert len(ds) == 2 and ds.flag.tolist() == [1, 0] and ds.cat2label[15] == 15
    np.random.seed(0)
    for i in range(len(ds)):
        d = ds[i]
        assert d['img'].dtype == torch.float32 and d['img'].shape[0] == 3
        assert d['img'].shape[1] % 32 == 0 and d['img'].shape[2] % 32 == 0
        assert d['gt_bboxes'].shape == (5, 8) and d['gt_bboxes'].dtype == torch.float32
        assert d['gt_labels'].dtype == torch.int64 and d['gt_labels'].min() >= 1
        m = d['img_meta']
        assert m['pad_shape'][:2] == tuple(d['img'].shape[1:]) and m['ori_shape'][2] == 3
    test_pipeline = [
        dict(type='LoadImageFromFile'),
        dict(type='MultiScaleFlipAug', img_scale=(256, 256), flip=False, transforms=[
            dict(type='RotateResize', keep_ratio=True), dict(type='RotateRandomFlip'), dict(type='Normalize', **IMG_NORM),
            dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])])]
    dt = dota.DotaDataset(ann_file=ann_file, img_prefix=tmp_path, pipeline=test_pipeline, test_mode=True)
    assert len(dt) == 3
    d = dt[0]
    assert isinstance(d['img'], list) and d['img'][0].shape == (3, 192, 256)
    assert d['img_meta'][0]['scale_factor'] == 2.0 and d['img_meta'][0]['flip'] is False


def test_rotate_pipeline_drops_images_whose_boxes_all_vanish(tmp_path):
    r = dict(img=IMG.copy(), img_shape=IMG.shape, gt_bboxes=np.array([[1, 1, 4, 1, 4, 3, 1, 3]], np.float32),
             gt_labels=np.array([1]), bbox_fields=['gt_bboxes'])
    np.random.seed(0)
    assert P.Compose([dict(type='PolyRandomRotate', rotate_ratio=0.0), dict(type='Pad', size_divisor=32)])(r) is None
    with pytest.raises(KeyError):
        P.build_pipeline(dict(type='NoSuchTransform'))
    np.random.seed(1)
    h = P.HSVAugment()(dict(img=IMG.copy() + 9))
    assert np.all(h['img'] == 9)
