"""CPU: the drop-in boundary -- the mirrors in orientedreppoints_amd/mmdet_ops + dota_devkit expose the
reference wrappers' names, parameter names and defaults (parsed with `ast` from the reference's own sources where they
lie; skipped when /root/reference is absent), and the reference's own Python files import and resolve their operators
through `compat.install_mmdet_aliases()`."""
import ast
import importlib
import importlib.util
import inspect
import os
import re
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _ref_ast(rel):
    return ast.parse(open(os.path.join(REF, rel)).read())


def _sig_from_ast(fn, drop_first=0):
    """[(name, default-source or None)] of an ast.FunctionDef (positional + keyword parameters)."""
    args = fn.args.args[drop_first:]
    defaults = [None] * (len(fn.args.args) - len(fn.args.defaults)) + list(fn.args.defaults)
    defaults = defaults[drop_first:]
    out = []
    for a, d in zip(args, defaults):
        out.append((a.arg, None if d is None else ast.literal_eval(d)))
    if fn.args.vararg:
        out.append(('*' + fn.args.vararg.arg, None))
    if fn.args.kwarg:
        out.append(('**' + fn.args.kwarg.arg, None))
    return out


def _sig_from_obj(obj, drop_first=0):
    out = []
    params = list(inspect.signature(obj).parameters.values())[drop_first:]
    for p in params:
        if p.kind == p.VAR_POSITIONAL:
            out.append(('*' + p.name, None))
        elif p.kind == p.VAR_KEYWORD:
            out.append(('**' + p.name, None))
        else:
            out.append((p.name, None if p.default is p.empty else p.default))
    return out


def _find(tree, name, cls=None):
    body = tree.body
    if cls is not None:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls][0].body
    return [n for n in body if isinstance(n, ast.FunctionDef) and n.name == name][0]


def test_function_wrappers_have_the_reference_signatures():
    from orientedreppoints_amd.mmdet_ops import chamfer_distance, iou_wrapper, minarea_rect, nms_wrapper
    t = _ref_ast("mmdet/ops/nms/nms_wrapper.py")
    for name in ("rnms", "soft_rnms"):
        assert _sig_from_obj(getattr(nms_wrapper, name)) == _sig_from_ast(_find(t, name)), name
    t = _ref_ast("mmdet/ops/iou/iou_wrapper.py")
    for name in ("convex_giou", "convex_iou", "convex_overlaps"):
        assert _sig_from_obj(getattr(iou_wrapper, name)) == _sig_from_ast(_find(t, name)), name
    t = _ref_ast("mmdet/ops/chamfer_distance.py")
    assert _sig_from_obj(chamfer_distance.ChamferDistance2D) == _sig_from_ast(_find(t, "ChamferDistance2D"))
    t = _ref_ast("mmdet/ops/minarearect/minarea_rect.py")
    assert _sig_from_obj(minarea_rect.minaerarect) == _sig_from_ast(_find(t, "minaerarect"))


def test_deform_conv_and_focal_modules_have_the_reference_signatures():
    dc = importlib.import_module("orientedreppoints_amd.mmdet_ops.deform_conv")
    t = _ref_ast("mmdet/ops/dcn/deform_conv.py")
    for cls in ("DeformConv", "ModulatedDeformConv", "DeformConvPack", "ModulatedDeformConvPack"):
        for meth in ("__init__", "forward"):
            assert _sig_from_obj(getattr(getattr(dc, cls), meth), 1) == _sig_from_ast(_find(t, meth, cls), 1), (cls, meth)
    for cls in ("DeformConvFunction", "ModulatedDeformConvFunction"):
        assert _sig_from_obj(getattr(dc, cls).forward, 1) == _sig_from_ast(_find(t, "forward", cls), 1), cls
    fl = importlib.import_module("orientedreppoints_amd.mmdet_ops.sigmoid_focal_loss")
    t = _ref_ast("mmdet/ops/sigmoid_focal_loss/sigmoid_focal_loss.py")
    assert _sig_from_obj(fl.SigmoidFocalLossFunction.forward, 1) == _sig_from_ast(_find(t, "forward", "SigmoidFocalLossFunction"), 1)
    assert _sig_from_obj(fl.SigmoidFocalLoss.__init__, 1) == _sig_from_ast(_find(t, "__init__", "SigmoidFocalLoss"), 1)


def test_cython_and_pybind_entry_points():
    from orientedreppoints_amd.dota_devkit import poly_nms_gpu as png
    from orientedreppoints_amd.mmdet_ops import box_iou_rotated, pointsJf

    def pyx_params(rel, fname):
        src = open(os.path.join(REF, rel)).read()
        m = re.search(r"def\s+%s\s*\((.*?)\):" % fname, src, re.S)
        params = re.sub(r"\[[^\]]*\]", "", m.group(1))            # drop the Cython buffer declarations [dtype, ndim=2]
        out = []
        for part in params.split(","):
            part = part.strip()
            if "=" in part:
                left, d = part.split("=")
                out.append((left.split()[-1], ast.literal_eval(d.strip())))
            else:
                out.append((part.split()[-1], None))
        return out
    assert _sig_from_obj(png.poly_gpu_nms) == pyx_params("DOTA_devkit/poly_nms_gpu/poly_nms.pyx", "poly_gpu_nms")
    assert _sig_from_obj(png.poly_overlaps) == pyx_params("DOTA_devkit/poly_nms_gpu/poly_overlaps.pyx", "poly_overlaps")
    # pybind functions: arity from the C++ declarations
    src = open(os.path.join(REF, "mmdet/ops/point_justify/src/points_justify.cpp")).read()
    assert len(re.search(r"int\s+pointsJf\s*\((.*?)\)", src, re.S).group(1).split(",")) == len(_sig_from_obj(pointsJf)) == 3
    src = open(os.path.join(REF, "mmdet/ops/box_iou_rotated/src/box_iou_rotated.h")).read()
    assert len(re.search(r"box_iou_rotated\s*\((.*?)\)", src, re.S).group(1).split(",")) == len(_sig_from_obj(box_iou_rotated)) == 2


def _load_ref(modname, rel):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    m.__package__ = modname.rsplit(".", 1)[0]
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def test_reference_python_imports_through_the_aliases():
    """bbox_nms.py, max_iou_assigner.py and iou_loss.py of the reference, loaded where they lie, bind their native
    operators to this package through install_mmdet_aliases() (the compiled extensions do not exist here)."""
    saved = {k: v for k, v in sys.modules.items() if k == "mmdet" or k.startswith("mmdet.") or k.startswith("DOTA_devkit")}
    try:
        for k in list(saved):
            del sys.modules[k]
        from orientedreppoints_amd.compat import install_mmdet_aliases
        from orientedreppoints_amd.mmdet_ops import iou_wrapper, nms_wrapper
        install_mmdet_aliases()
        bn = _load_ref("mmdet.core.post_processing.bbox_nms", "mmdet/core/post_processing/bbox_nms.py")
        assert bn.nms_wrapper is nms_wrapper and bn.nms_wrapper.rnms is nms_wrapper.rnms
        # max_iou_assigner: stub only its non-operator siblings
        class NiceRepr(object):
            pass
        for name, attrs in (("mmdet.utils", dict(util_mixins=types.SimpleNamespace(NiceRepr=NiceRepr))),
                            ("mmdet.utils.util_mixins", dict(NiceRepr=NiceRepr)), ("mmdet.core", {}),
                            ("mmdet.core.bbox", {}), ("mmdet.core.bbox.assigners", {})):
            m = types.ModuleType(name); m.__path__ = []; m.__dict__.update(attrs); sys.modules[name] = m
        _load_ref("mmdet.core.bbox.assigners.assign_result", "mmdet/core/bbox/assigners/assign_result.py")
        _load_ref("mmdet.core.bbox.assigners.base_assigner", "mmdet/core/bbox/assigners/base_assigner.py")
        mia = _load_ref("mmdet.core.bbox.assigners.max_iou_assigner", "mmdet/core/bbox/assigners/max_iou_assigner.py")
        assert mia.convex_overlaps is iou_wrapper.convex_overlaps
        for name, attrs in (("mmdet.models", {}), ("mmdet.models.losses", {}),
                            ("mmdet.models.registry", dict(LOSSES=types.SimpleNamespace(register_module=lambda c=None: c if c else (lambda k: k))))):
            m = types.ModuleType(name); m.__path__ = []; m.__dict__.update(attrs); sys.modules[name] = m
        sys.modules["mmdet.core"].bbox_overlaps = None
        _load_ref("mmdet.models.losses.utils", "mmdet/models/losses/utils.py")
        il = _load_ref("mmdet.models.losses.iou_loss", "mmdet/models/losses/iou_loss.py")
        assert il.convex_giou is iou_wrapper.convex_giou
        import DOTA_devkit.poly_nms_gpu.poly_nms as pn
        assert callable(pn.poly_gpu_nms)
        from mmdet.ops import DeformConv, minaerarect, sigmoid_focal_loss  # noqa: F401
    finally:
        for k in [k for k in sys.modules if k == "mmdet" or k.startswith("mmdet.") or k.startswith("DOTA_devkit")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_box_iou_rotated_cpu_branch_matches_reference_golden(golden_dir):
    """box_iou_rotated.h:20-33 dispatches CPU tensors to box_iou_rotated_cpu: the mirror does too (host-compiled same
    source as the device kernel) and reproduces the reference-generated golden exactly."""
    import numpy as np
    import torch
    from orientedreppoints_amd.mmdet_ops import box_iou_rotated
    g = np.load(os.path.join(golden_dir, "box_iou_rotated.npz"))
    got = box_iou_rotated(torch.from_numpy(g["a"]), torch.from_numpy(g["b"]))
    assert not got.is_cuda and got.shape == g["iou"].shape
    assert np.max(np.abs(got.numpy() - g["iou"])) <= 1e-6
    assert box_iou_rotated(torch.zeros(0, 5), torch.from_numpy(g["b"])).shape == (0, g["b"].shape[0])
