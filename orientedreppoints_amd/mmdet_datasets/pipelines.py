"""Pipeline transforms the DOTA configs name (configs/dota/*.py `train_pipeline` / `test_pipeline`): same class names,
constructor arguments, `results` dict keys and box arithmetic as mmdet/datasets/pipelines/{transforms.py:43-270,
poly_transforms.py:15-546, formating.py, loading.py, test_aug.py, compose.py}.  Random draws use the same numpy / random
calls in the same order as the reference, so a seeded pipeline makes the same decisions."""
import collections
import os
import random

import numpy as np
import torch

from . import imops
from .geometry import box_points, min_area_rect, poly2rbox, rbox2poly


class _Registry(object):
    def __init__(self, name):
        self.name, self.module_dict = name, {}

    def register_module(self, cls=None):
        if cls is None:
            return self.register_module
        self.module_dict[cls.__name__] = cls
        return cls

    def get(self, key):
        return self.module_dict.get(key)


PIPELINES = _Registry('pipeline')


def build_pipeline(cfg):
    args = dict(cfg)
    t = args.pop('type')
    cls = PIPELINES.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError('%s is not in the pipeline registry' % t)
    return cls(**args)


@PIPELINES.register_module
class Compose(object):
    def __init__(self, transforms):
        self.transforms = [build_pipeline(t) if isinstance(t, dict) else t for t in transforms]

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
        return data


@PIPELINES.register_module
class LoadImageFromFile(object):
    """Reads .npy arrays and binary PPM / PGM (no image codec library in this environment); loading.py:12-39."""

    def __init__(self, to_float32=False, color_type='color'):
        self.to_float32 = to_float32

    @staticmethod
    def _read(path):
        if path.endswith('.npy'):
            return np.load(path)
        with open(path, 'rb') as f:
            magic = f.readline().strip()
            if magic not in (b'P5', b'P6'):
                raise IOError('LoadImageFromFile: only .npy and binary PPM/PGM are readable here (%s)' % path)
            line = f.readline()
            while line.startswith(b'#'):
                line = f.readline()
            w, h = map(int, line.split())
            f.readline()
            data = np.frombuffer(f.read(), dtype=np.uint8)
        img = data.reshape(h, w, 3 if magic == b'P6' else 1)
        return img[..., ::-1].copy() if magic == b'P6' else np.repeat(img, 3, 2)        # BGR like cv2.imread

    def __call__(self, results):
        filename = os.path.join(results['img_prefix'], results['img_info']['filename']) \
            if results.get('img_prefix') is not None else results['img_info']['filename']
        img = self._read(filename)
        if self.to_float32:
            img = img.astype(np.float32)
        results['filename'] = filename
        results['img'] = img
        results['img_shape'] = img.shape
        results['ori_shape'] = img.shape
        return results


@PIPELINES.register_module
class LoadAnnotations(object):
    def __init__(self, with_bbox=True, with_label=True, with_mask=False, with_seg=False, poly2mask=True):
        self.with_bbox, self.with_label = with_bbox, with_label

    def __call__(self, results):
        ann = results['ann_info']
        if self.with_bbox:
            results['gt_bboxes'] = ann['bboxes']
            if ann.get('bboxes_ignore') is not None:
                results['gt_bboxes_ignore'] = ann['bboxes_ignore']
                results['bbox_fields'].append('gt_bboxes_ignore')
            results['bbox_fields'].append('gt_bboxes')
        if self.with_label:
            results['gt_labels'] = ann['labels']
        return results


@PIPELINES.register_module
class CorrectBox(object):
    """gt quads -> their minimum-area rotated rectangles; refine_rbbox keeps the corner nearest to the original first
    corner in front (transforms.py:43-83)."""

    def __init__(self, correct_rbbox=True, refine_rbbox=False):
        self.correct_rbbox, self.refine_rbbox = correct_rbbox, refine_rbbox

    def _correct_rbbox(self, gt_rbboxes_points, refine_rbbox=False):
        out = []
        for rbbox_points in gt_rbboxes_points:
            pts = rbbox_points.astype(np.int64).reshape(4, 2)
            rect_pts = box_points(min_area_rect(pts)).reshape(-1)
            if refine_rbbox:
                min_dist, index = 1e8, 0
                for i, p in enumerate(rect_pts.reshape(4, 2)):
                    dist = np.sqrt((pts[0][0] - p[0]) ** 2 + (pts[0][1] - p[1]) ** 2)
                    if dist <= min_dist:
                        min_dist, index = dist, i
                rect_pts = np.array([rect_pts[2 * ((index + k) % 4) + c] for k in range(4) for c in (0, 1)])
            out.append(rect_pts)
        return np.array(out)

    def __call__(self, results):
        if self.correct_rbbox:
            many = results if isinstance(results, list) else [results]
            for r in many:
                r['gt_bboxes'] = self._correct_rbbox(r['gt_bboxes'], self.refine_rbbox).astype(np.float32).reshape(-1, 8)
        return results


CorrectRBBox = PIPELINES.register_module(type('CorrectRBBox', (CorrectBox,), {}))


@PIPELINES.register_module
class RotateResize(object):
    """transforms.py:85-200 (and poly_transforms.py PolyResize, which adds `interpolation`)."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True, clamp_rbbox=True,
                 interpolation='bilinear'):
        self.clamp_rbbox, self.interpolation = clamp_rbbox, interpolation
        self.img_scale = None if img_scale is None else (img_scale if isinstance(img_scale, list) else [img_scale])
        if self.img_scale is not None:
            self.img_scale = [tuple(s) for s in self.img_scale]
        if ratio_range is not None:
            assert len(self.img_scale) == 1
        else:
            assert multiscale_mode in ['value', 'range']
        self.multiscale_mode, self.ratio_range, self.keep_ratio = multiscale_mode, ratio_range, keep_ratio

    @staticmethod
    def random_select(img_scales):
        scale_idx = np.random.randint(len(img_scales))
        return img_scales[scale_idx], scale_idx

    @staticmethod
    def random_sample(img_scales):
        assert len(img_scales) == 2
        img_scale_long = [max(s) for s in img_scales]
        img_scale_short = [min(s) for s in img_scales]
        long_edge = np.random.randint(min(img_scale_long), max(img_scale_long) + 1)
        short_edge = np.random.randint(min(img_scale_short), max(img_scale_short) + 1)
        return (long_edge, short_edge), None

    @staticmethod
    def random_sample_ratio(img_scale, ratio_range):
        min_ratio, max_ratio = ratio_range
        assert min_ratio <= max_ratio
        ratio = np.random.random_sample() * (max_ratio - min_ratio) + min_ratio
        return (int(img_scale[0] * ratio), int(img_scale[1] * ratio)), None

    def _random_scale(self, results):
        if self.ratio_range is not None:
            scale, scale_idx = self.random_sample_ratio(self.img_scale[0], self.ratio_range)
        elif len(self.img_scale) == 1:
            scale, scale_idx = self.img_scale[0], 0
        elif self.multiscale_mode == 'range':
            scale, scale_idx = self.random_sample(self.img_scale)
        else:
            scale, scale_idx = self.random_select(self.img_scale)
        results['scale'], results['scale_idx'] = scale, scale_idx

    def _resize_img(self, results):
        if self.keep_ratio:
            img, scale_factor = imops.imrescale(results['img'], results['scale'], return_scale=True,
                                                interpolation=self.interpolation)
        else:
            img, w_scale, h_scale = imops.imresize(results['img'], results['scale'], return_scale=True,
                                                   interpolation=self.interpolation)
            scale_factor = np.array([w_scale, h_scale, w_scale, h_scale], dtype=np.float32)
        results['img'] = img
        results['img_shape'] = img.shape
        results['pad_shape'] = img.shape
        results['scale_factor'] = scale_factor
        results['keep_ratio'] = self.keep_ratio

    def _resize_bboxes(self, results, clamp_rbbox=True):
        img_shape = results['img_shape']
        for key in results.get('bbox_fields', []):
            bboxes = results[key] * results['scale_factor']
            if clamp_rbbox:
                bboxes[:, 0::2] = np.clip(bboxes[:, 0::2], 0, img_shape[1] - 1)
                bboxes[:, 1::2] = np.clip(bboxes[:, 1::2], 0, img_shape[0] - 1)
            results[key] = bboxes

    def _one(self, results):
        if 'scale' not in results:
            self._random_scale(results)
        self._resize_img(results)
        self._resize_bboxes(results, self.clamp_rbbox)
        return results

    def __call__(self, results):
        if isinstance(results, list):
            return [self._one(r) for r in results]
        return self._one(results)


@PIPELINES.register_module
class PolyResize(RotateResize):
    """poly_transforms.py:86-247: RotateResize plus the test-time random edge re-draw when `scale` is preset (:207-216)."""

    def _one(self, results):
        if 'scale' not in results:
            self._random_scale(results)
        else:
            assert len(results['scale']) == 2
            edge1 = np.random.randint(min(results['scale']), max(results['scale']) + 1)
            edge2 = np.random.randint(min(results['scale']), max(results['scale']) + 1)
            results['scale'] = (max(edge1, edge2) + 1, min(edge1, edge2))
        self._resize_img(results)
        self._resize_bboxes(results, self.clamp_rbbox)
        return results

    def __call__(self, results):
        if isinstance(results, list):            # mosaic lists keep a preset scale (multi_img_call :221-228)
            return [RotateResize._one(self, r) for r in results]
        return self._one(results)


def rbbox_flip(rbboxes, img_shape, direction):
    """Flip [..., 8k] quads (transforms.py:224-248 / poly_transforms.py:273-297)."""
    assert rbboxes.shape[-1] % 8 == 0
    flipped = rbboxes.copy()
    if direction == 'horizontal':
        w = img_shape[1]
        for k in (0, 2, 4, 6):
            flipped[..., k::8] = w - rbboxes[..., k::8] - 1
    elif direction == 'vertical':
        h = img_shape[0]
        for k in (1, 3, 5, 7):
            flipped[..., k::8] = h - rbboxes[..., k::8] - 1
    else:
        raise ValueError('Invalid flipping direction "{}"'.format(direction))
    return flipped


def rbbox_mapping_back(bboxes, img_shape, scale_factor, flip, filp_direction):
    """Test-time boxes [n, 8k] (torch) back to the original image (core/bbox/transforms.py:273-301)."""
    if flip:
        flipped = bboxes.clone()
        if filp_direction == 'horizontal':
            flipped[:, 0::2] = img_shape[1] - bboxes[:, 0::2] - 1
        else:
            flipped[:, 1::2] = img_shape[0] - bboxes[:, 1::2] - 1
        bboxes = flipped
    return bboxes / scale_factor


@PIPELINES.register_module
class RotateRandomFlip(object):
    """transforms.py:203-270."""

    def __init__(self, flip_ratio=None, direction=['horizontal']):
        self.flip_ratio, self.direction = flip_ratio, direction
        if flip_ratio is not None:
            assert 0 <= flip_ratio <= 1
        for d in self.direction:
            assert d in ['horizontal', 'vertical']

    rbbox_flip = staticmethod(rbbox_flip)

    def _draw_direction(self, results):
        if 'flip_direction' not in results:
            results['flip_direction'] = np.random.choice(self.direction, 1)

    def _one(self, results):
        if 'flip' not in results:
            results['flip'] = True if np.random.rand() < self.flip_ratio else False
        self._draw_direction(results)
        if results['flip']:
            d = results['flip_direction']
            d = d if isinstance(d, str) else str(np.asarray(d).reshape(-1)[0])
            results['img'] = imops.imflip(results['img'], direction=d)
            for key in results.get('bbox_fields', []):
                results[key] = rbbox_flip(results[key], results['img_shape'], d)
        return results

    def __call__(self, results):
        if isinstance(results, list):
            return [self._one(r) for r in results]
        return self._one(results)


@PIPELINES.register_module
class PolyRandomFlip(RotateRandomFlip):
    """poly_transforms.py:249-345: the direction is re-drawn on every call with random.sample."""

    def __init__(self, flip_ratio=None, direction=['horizontal', 'vertical']):
        super(PolyRandomFlip, self).__init__(flip_ratio, direction)

    def _draw_direction(self, results):
        results['flip_direction'] = random.sample(self.direction, 1)[0]


@PIPELINES.register_module
class PolyRandomRotate(object):
    """poly_transforms.py:348-546: rotate the image about its centre, transform the quad corners with the same matrix,
    re-fit (poly2rbox), drop boxes whose centre left the image or that became smaller than 5 px, back to quads.
    NB the reference passes the drawn angle (degrees for cv2) to np.cos / np.sin as radians when it sizes the
    auto_bound canvas (:432-437); kept."""

    def __init__(self, rotate_ratio=0.5, angles_range=180, auto_bound=False):
        self.rotate_ratio, self.auto_bound, self.angles_range = rotate_ratio, auto_bound, angles_range
        self.discrete_range = [90, 180, -90, -180]

    @property
    def is_rotate(self):
        return np.random.rand() < self.rotate_ratio

    def create_rotation_matrix(self, center, angle, bound_h, bound_w, offset=0):
        center = (center[0] + offset, center[1] + offset)
        rm = imops.rotation_matrix_2d(center, angle, 1)
        if self.auto_bound:
            c = np.asarray(center, np.float64) + offset
            rot_im_center = rm[:, :2].dot(c) + rm[:, 2]
            new_center = np.array([bound_w / 2, bound_h / 2]) + offset - rot_im_center
            rm[:, 2] += new_center
        return rm

    def apply_coords(self, coords):
        if len(coords) == 0:
            return coords
        coords = np.asarray(coords, dtype=float)
        return coords.dot(self.rm_coords[:, :2].T) + self.rm_coords[:, 2]

    @staticmethod
    def filter_border(bboxes, h, w):
        x_ctr, y_ctr, w_bbox, h_bbox = bboxes[:, 0], bboxes[:, 1], bboxes[:, 2], bboxes[:, 3]
        return (x_ctr > 0) & (x_ctr < w) & (y_ctr > 0) & (y_ctr < h) & (w_bbox > 5) & (h_bbox > 5)

    def _one(self, results):
        if not self.is_rotate:
            results['rotate'] = False
            angle = 0
        else:
            angle = random.uniform(-self.angles_range, self.angles_range)
            results['rotate'] = True
        h, w, c = results['img_shape']
        results['rotate_angle'] = angle
        image_center = np.array((w / 2, h / 2))
        abs_cos, abs_sin = abs(np.cos(angle)), abs(np.sin(angle))
        if self.auto_bound:
            bound_w, bound_h = np.rint([h * abs_sin + w * abs_cos, h * abs_cos + w * abs_sin]).astype(int)
        else:
            bound_w, bound_h = w, h
        self.rm_coords = self.create_rotation_matrix(image_center, angle, bound_h, bound_w)
        self.rm_image = self.create_rotation_matrix(image_center, angle, bound_h, bound_w, offset=-0.5)
        results['img'] = imops.warp_affine(results['img'], self.rm_image, (bound_w, bound_h))
        results['img_shape'] = (bound_h, bound_w, c)
        gt_bboxes = results.get('gt_bboxes', np.zeros((0, 8), np.float32))
        labels = results.get('gt_labels', np.zeros((0,), np.int64))
        polys = self.apply_coords(gt_bboxes.reshape(-1, 2)).reshape(-1, 8)
        rb = poly2rbox(polys) if len(polys) else np.zeros((0, 5))
        keep = self.filter_border(rb, bound_h, bound_w) if len(rb) else np.zeros((0,), bool)
        rb, labels = rb[keep, :], labels[keep]
        if len(rb) == 0:
            return None
        results['gt_bboxes'] = rbox2poly(rb).astype(np.float32)
        results['gt_labels'] = labels
        return results

    def __call__(self, results):
        if isinstance(results, list):
            return [self._one(r) for r in results]
        return self._one(results)


@PIPELINES.register_module
class HSVAugment(object):
    """transforms.py:1156-1212.  The reference converts into a temporary (`dst=img.astype(np.float32)`), so the image it
    returns is the one it was given; what it does do is consume three uniform draws per image, which is kept so that a
    seeded pipeline stays in step."""

    def __init__(self, hgain=0.015, sgain=0.7, vgain=0.4):
        self.hgain, self.sgain, self.vgain = hgain, sgain, vgain

    def __call__(self, results):
        for r in (results if isinstance(results, list) else [results]):
            np.random.uniform(-1, 1, 3)
        return results


@PIPELINES.register_module
class Normalize(object):
    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, np.float32), np.array(std, np.float32), to_rgb

    def __call__(self, results):
        results['img'] = imops.imnormalize(results['img'], self.mean, self.std, self.to_rgb)
        results['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results


@PIPELINES.register_module
class Pad(object):
    def __init__(self, size=None, size_divisor=None, pad_val=0):
        self.size, self.size_divisor, self.pad_val = size, size_divisor, pad_val
        assert size is not None or size_divisor is not None
        assert size is None or size_divisor is None

    def __call__(self, results):
        img = imops.impad(results['img'], self.size, self.pad_val) if self.size is not None else \
            imops.impad_to_multiple(results['img'], self.size_divisor, self.pad_val)
        results['img'] = img
        results['pad_shape'] = img.shape
        results['pad_fixed_size'], results['pad_size_divisor'] = self.size, self.size_divisor
        return results


@PIPELINES.register_module
class ImageToTensor(object):
    def __init__(self, keys):
        self.keys = keys

    def __call__(self, results):
        for key in self.keys:
            results[key] = torch.from_numpy(np.ascontiguousarray(results[key].transpose(2, 0, 1)))
        return results


@PIPELINES.register_module
class DefaultFormatBundle(object):
    """formating.py:124-163 without mmcv's DataContainer: tensors go in as plain torch tensors."""

    def __call__(self, results):
        if 'img' in results:
            img = results['img']
            if img.ndim < 3:
                img = np.expand_dims(img, -1)
            results['img'] = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        for key in ['proposals', 'gt_bboxes', 'gt_bboxes_ignore', 'gt_labels']:
            if key in results:
                results[key] = torch.from_numpy(np.ascontiguousarray(results[key]))
        return results


@PIPELINES.register_module
class Collect(object):
    def __init__(self, keys, meta_keys=('filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                                        'img_norm_cfg')):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = {'img_meta': {k: results[k] for k in self.meta_keys if k in results}}
        for key in self.keys:
            data[key] = results[key]
        return data


@PIPELINES.register_module
class MultiScaleFlipAug(object):
    """test_aug.py:7-38."""

    def __init__(self, transforms, img_scale, flip=False):
        self.transforms = Compose(transforms)
        self.img_scale = img_scale if isinstance(img_scale, list) else [img_scale]
        self.flip = flip

    def __call__(self, results):
        aug_data = []
        flip_aug = [False, True] if self.flip else [False]
        for scale in self.img_scale:
            for flip in flip_aug:
                _results = results.copy()
                _results['scale'] = scale
                _results['flip'] = flip
                aug_data.append(self.transforms(_results))
        aug_data_dict = collections.defaultdict(list)
        for data in aug_data:
            for key, val in data.items():
                aug_data_dict[key].append(val)
        return dict(aug_data_dict)
