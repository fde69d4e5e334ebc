"""Test-time augmentation glue of the detector (reference orientedreppoints_detector.py:49-144): flip / rescale mapping of
the views' candidates and their concatenation, against a numpy restatement of the reference's index arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from orientedreppoints_amd.mmdet_models.detector import OrientedRepPointsDetector


def _bare_detector():
    det = OrientedRepPointsDetector.__new__(OrientedRepPointsDetector)
    nn.Module.__init__(det)
    return det


def _flip_np(b, img_shape, direction):
    """orientedreppoints_detector.py:49-73, index by index."""
    out = b.copy()
    if direction == 'horizontal':
        for k in (0, 2, 4, 6):
            out[..., k::8] = np.float32(img_shape[1]) - b[..., k::8] - np.float32(1)
    else:
        for k in (1, 3, 5, 7):
            out[..., k::8] = np.float32(img_shape[0]) - b[..., k::8] - np.float32(1)
    return out


@pytest.mark.parametrize('k', [1, 3])
def test_rbbox_flip_matches_the_reference_arithmetic(k):
    rng = np.random.RandomState(0)
    b = (rng.rand(7, 8 * k) * 300).astype(np.float32)
    t = torch.from_numpy(b)
    for direction in ('horizontal', 'vertical'):
        got = OrientedRepPointsDetector.rbbox_flip(t, (200, 320, 3), direction).numpy()
        assert np.array_equal(got, _flip_np(b, (200, 320, 3), direction))
    assert np.array_equal(t.numpy(), b)                     # the input is not modified
    with pytest.raises(ValueError):
        OrientedRepPointsDetector.rbbox_flip(t, (200, 320, 3), 'diagonal')
    with pytest.raises(AssertionError):
        OrientedRepPointsDetector.rbbox_flip(t[:, :7], (200, 320, 3))


def test_merge_aug_results_maps_every_view_back():
    det = _bare_detector()
    rng = np.random.RandomState(1)
    views = [(rng.rand(5, 8).astype(np.float32) * 256, dict(img_shape=(256, 256, 3), scale_factor=1.0, flip=False)),
             (rng.rand(0, 8).astype(np.float32), dict(img_shape=(256, 256, 3), scale_factor=1.0, flip=True)),
             (rng.rand(4, 8).astype(np.float32) * 128, dict(img_shape=(128, 128, 3), scale_factor=0.5, flip=True)),
             (rng.rand(3, 8).astype(np.float32) * 384, dict(img_shape=(384, 384, 3), scale_factor=1.5, flip=False))]
    scores = [rng.rand(v[0].shape[0], 16).astype(np.float32) for v in views]
    boxes, sc = det.merge_aug_results([torch.from_numpy(v[0]) for v in views], [torch.from_numpy(s) for s in scores],
                                      [[v[1]] for v in views])
    want = np.concatenate([(_flip_np(b, m['img_shape'], 'horizontal') if m['flip'] else b) / np.float32(m['scale_factor'])
                           for b, m in views])
    assert np.array_equal(boxes.numpy(), want)
    assert np.array_equal(sc.numpy(), np.concatenate(scores))
    only_boxes = det.merge_aug_results([torch.from_numpy(v[0]) for v in views], None, [[v[1]] for v in views])
    assert np.array_equal(only_boxes.numpy(), want)


def test_forward_test_dispatch():
    det = _bare_detector()
    calls = []
    det.simple_test = lambda img, metas, **kw: calls.append(('simple', img, metas, kw)) or 'S'
    det.aug_test = lambda imgs, metas, **kw: calls.append(('aug', imgs, metas, kw)) or 'A'
    x = torch.zeros(1, 3, 8, 8)
    assert det.forward_test(x, [dict()]) == 'S'
    assert det.forward_test([x], [[dict()]], rescale=True) == 'S' and calls[-1][3] == dict(rescale=True)
    assert det.forward_test([x, x], [[dict()], [dict()]]) == 'A' and len(calls[-1][1]) == 2
    with pytest.raises(ValueError):
        det.forward_test([x, x], [[dict()]])
    with pytest.raises(AssertionError):
        det.forward_test([torch.zeros(2, 3, 8, 8)] * 2, [[dict()] * 2] * 2)       # one image per GPU at test
