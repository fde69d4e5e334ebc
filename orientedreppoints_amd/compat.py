"""`install_mmdet_aliases()`: re-point the reference's import paths to this package.

After the call, `from mmdet.ops.nms import nms_wrapper`, `from mmdet.ops.iou import convex_giou`,
`from mmdet.ops.minarearect import minaerarect`, `from mmdet.ops import DeformConv, sigmoid_focal_loss`,
`from DOTA_devkit.poly_nms_gpu.poly_nms import poly_gpu_nms` ... resolve to the MI355X operators, so the reference's own
Python files (bbox_nms.py, max_iou_assigner.py, iou_loss.py, the head) run on liborp_hip.so unchanged.  Parent packages
(`mmdet`, `mmdet.ops`, `DOTA_devkit`) are created as empty namespaces only when they are not importable already.
"""
import importlib
import sys
import types


def _namespace(name):
    m = sys.modules.get(name)
    if m is None:
        try:
            m = importlib.import_module(name)
        except Exception:                               # not installed (mmcv / compiled extensions missing): a stand-in
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
            if '.' in name:
                parent, leaf = name.rsplit('.', 1)
                setattr(_namespace(parent), leaf, m)
    return m


def _alias(name, module):
    sys.modules[name] = module
    parent, leaf = name.rsplit('.', 1)
    setattr(_namespace(parent), leaf, module)


def install_mmdet_aliases():
    from . import mmdet_ops as ops
    from .dota_devkit import poly_nms_gpu as png
    from .mmdet_ops import (box_iou_rotated, chamfer_distance, deform_conv, iou_wrapper, minarea_rect, nms_wrapper,
                            point_justify, sigmoid_focal_loss)
    mods = dict(deform_conv=importlib.import_module(ops.__name__ + '.deform_conv'),
                sigmoid_focal_loss=importlib.import_module(ops.__name__ + '.sigmoid_focal_loss'),
                box_iou_rotated=importlib.import_module(ops.__name__ + '.box_iou_rotated'))
    del box_iou_rotated, deform_conv, sigmoid_focal_loss  # the package re-exports functions of these names
    ops_ns = _namespace('mmdet.ops')
    for name in ops.__dict__:
        if not name.startswith('_'):
            setattr(ops_ns, name, getattr(ops, name))    # from mmdet.ops import DeformConv, sigmoid_focal_loss, ...
    # sub-packages of mmdet.ops: each exposes what the reference's __init__ of that name exports
    nms_pkg = types.ModuleType('mmdet.ops.nms')
    nms_pkg.__path__ = []
    for k in ('rnms', 'soft_rnms', 'rnms_cuda', 'poly_nms_gpu'):
        setattr(nms_pkg, k, getattr(nms_wrapper, k))
    nms_pkg.nms_wrapper = nms_wrapper
    _alias('mmdet.ops.nms', nms_pkg)
    _alias('mmdet.ops.nms.nms_wrapper', nms_wrapper)
    _alias('mmdet.ops.iou', iou_wrapper)
    _alias('mmdet.ops.minarearect', minarea_rect)
    _alias('mmdet.ops.chamfer_distance', chamfer_distance)
    _alias('mmdet.ops.point_justify', point_justify)
    _alias('mmdet.ops.dcn', mods['deform_conv'])
    _alias('mmdet.ops.sigmoid_focal_loss', mods['sigmoid_focal_loss'])
    _alias('mmdet.ops.box_iou_rotated', mods['box_iou_rotated'])
    _namespace('DOTA_devkit.poly_nms_gpu')
    _alias('DOTA_devkit.poly_nms_gpu.poly_nms', png)
    _alias('DOTA_devkit.poly_nms_gpu.poly_overlaps', png)
    _alias('DOTA_devkit.poly_nms_gpu.nms_wrapper', png)
    return ops_ns
