"""Mirror of mmdet/ops/iou/iou_wrapper.py:12-27 (convex_iou / convex_overlaps; convex_giou is added with its kernel)
and the `convex_iou_cuda.convex_iou` extension function (mmdet/ops/iou/src/convex_iou_cuda.cpp:8-16)."""
import torch

from .. import _lib


class _ConvexIouCuda(object):
    @staticmethod
    def convex_iou(pred, target):
        """pred [N,18], target [K,8] CUDA f32 -> flat [N*K]; empty -> empty CPU float tensor."""
        _lib.require_cuda(pred, "pred")
        _lib.require_cuda(target, "target")
        if pred.numel() == 0 or target.numel() == 0:
            return torch.empty((0,), dtype=torch.float32, device="cpu")
        L = _lib.lib()
        p = pred.detach().float().reshape(-1, 18).contiguous()
        g = target.detach().float().reshape(-1, 8).contiguous()
        n, k = p.size(0), g.size(0)
        out = torch.empty((n * k,), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            rc = L.orp_convex_iou(_lib.ptr(p), n, _lib.ptr(g), k, _lib.ptr(out), _lib.stream_of(p))
        _lib.check(rc, "orp_convex_iou")
        return out


convex_iou_cuda = _ConvexIouCuda()


class _ConvexGiouCuda(object):
    @staticmethod
    def convex_giou(pred, target):
        """pred [P,18], target [P,8] CUDA f32 (aligned pairs) -> flat [P*19] = 18 grads + giou per row
        (mmdet/ops/iou/src/convex_giou_cuda.cpp, convex_giou_kernel.cu:806-868)."""
        _lib.require_cuda(pred, "pred")
        _lib.require_cuda(target, "target")
        if pred.numel() == 0 or target.numel() == 0:
            return torch.empty((0,), dtype=torch.float32, device="cpu")
        p = pred.detach().float().reshape(-1, 18).contiguous()
        g = target.detach().float().reshape(-1, 8).contiguous()
        assert p.size(0) == g.size(0), "ex_boxes must equal to gt_boxes"
        out = torch.empty((p.size(0) * 19,), dtype=torch.float32, device=p.device)
        with torch.cuda.device(p.device):
            rc = _lib.lib().orp_convex_giou(_lib.ptr(p), _lib.ptr(g), p.size(0), _lib.ptr(out), _lib.stream_of(p))
        _lib.check(rc, "orp_convex_giou")
        return out


convex_giou_cuda = _ConvexGiouCuda()


def convex_giou(pred, target):
    convex_giou_grad = convex_giou_cuda.convex_giou(pred, target)
    convex_giou_grad = convex_giou_grad.reshape(-1, 19)
    convex_giou = convex_giou_grad[:, -1]
    points_grad = convex_giou_grad[:, 0:-1]
    return convex_giou, points_grad


def convex_iou(pred, target):
    ex_num, gt_num = pred.size(0), target.size(0)
    convex_ious = convex_iou_cuda.convex_iou(pred, target)
    convex_ious = convex_ious.reshape(ex_num, gt_num)
    return convex_ious


def convex_overlaps(gt_rbboxes, points):
    overlaps = convex_iou(points, gt_rbboxes)
    overlaps = overlaps.transpose(1, 0)
    return overlaps
