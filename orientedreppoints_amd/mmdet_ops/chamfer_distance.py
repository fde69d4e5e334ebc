"""Mirror of mmdet/ops/chamfer_distance.py:6-22 and mmdet/ops/chamfer_2d/dist_chamfer_2d.py (ChamferFunction2D,
Chamfer2D) on the MI355X HIP library."""
import torch
from torch import nn
from torch.autograd import Function

from .. import _lib


class _Chamfer2dExt(object):
    """Stands in for the pybind module `chamfer_2d` (forward / backward fill caller-allocated tensors)."""

    @staticmethod
    def forward(xyz1, xyz2, dist1, dist2, idx1, idx2):
        for t, n in ((xyz1, "xyz1"), (xyz2, "xyz2"), (dist1, "dist1"), (dist2, "dist2"), (idx1, "idx1"), (idx2, "idx2")):
            _lib.require_cuda(t, n)
        b, n, _ = xyz1.size()
        m = xyz2.size(1)
        with torch.cuda.device(xyz1.device):
            rc = _lib.lib().orp_chamfer2d_forward(_lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(dist1),
                                                  _lib.ptr(dist2), _lib.ptr(idx1), _lib.ptr(idx2),
                                                  _lib.stream_of(xyz1))
        _lib.check(rc, "orp_chamfer2d_forward")
        return 1

    @staticmethod
    def backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2):
        b, n, _ = xyz1.size()
        m = xyz2.size(1)
        with torch.cuda.device(xyz1.device):
            rc = _lib.lib().orp_chamfer2d_backward(_lib.ptr(xyz1), _lib.ptr(xyz2), b, n, m, _lib.ptr(graddist1),
                                                   _lib.ptr(graddist2), _lib.ptr(idx1), _lib.ptr(idx2),
                                                   _lib.ptr(gradxyz1), _lib.ptr(gradxyz2), _lib.stream_of(xyz1))
        _lib.check(rc, "orp_chamfer2d_backward")
        return 1


chamfer_2d = _Chamfer2dExt()


class ChamferFunction2D(Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        batchsize, n, _ = xyz1.size()
        _, m, _ = xyz2.size()
        device = xyz1.device
        dist1 = torch.zeros(batchsize, n, device=device)
        dist2 = torch.zeros(batchsize, m, device=device)
        idx1 = torch.zeros(batchsize, n, dtype=torch.int32, device=device)
        idx2 = torch.zeros(batchsize, m, dtype=torch.int32, device=device)
        chamfer_2d.forward(xyz1, xyz2, dist1, dist2, idx1, idx2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, graddist1, graddist2, gradidx1, gradidx2):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        graddist1 = graddist1.contiguous()
        graddist2 = graddist2.contiguous()
        gradxyz1 = torch.zeros(xyz1.size(), device=graddist1.device)
        gradxyz2 = torch.zeros(xyz2.size(), device=graddist1.device)
        chamfer_2d.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
        return gradxyz1, gradxyz2


class Chamfer2D(nn.Module):
    def forward(self, input1, input2):
        input1 = input1.contiguous().float()
        input2 = input2.contiguous().float()
        return ChamferFunction2D.apply(input1, input2)


def ChamferDistance2D(point_set_1, point_set_2, distance_weight=0.05, eps=1e-12, use_cuda=True):
    chamfer = Chamfer2D()
    assert point_set_1.dim() == point_set_2.dim()
    assert point_set_1.shape[-1] == point_set_2.shape[-1]
    if point_set_1.dim() <= 3:
        if use_cuda:
            dist1, dist2, _, _ = chamfer(point_set_1, point_set_2)
            dist1 = torch.sqrt(torch.clamp(dist1, eps))
            dist2 = torch.sqrt(torch.clamp(dist2, eps))
            dist = (dist1.mean(-1) + dist2.mean(-1)) / 2.0
        else:
            dist = chamfer(point_set_1, point_set_2)
        return dist * distance_weight
