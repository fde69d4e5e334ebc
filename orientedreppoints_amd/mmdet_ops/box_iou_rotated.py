"""Mirror of mmdet/ops/box_iou_rotated (`box_iou_rotated(boxes1[N,5], boxes2[K,5]) -> [N,K]`, box_iou_rotated.h:20-38;
theta in radians)."""
import torch

from .. import _lib


def box_iou_rotated(boxes1, boxes2):
    _lib.require_cuda(boxes1, "boxes1")
    _lib.require_cuda(boxes2, "boxes2")
    a = boxes1.detach().float().reshape(-1, 5).contiguous()
    b = boxes2.detach().float().reshape(-1, 5).contiguous()
    out = torch.empty((a.size(0), b.size(0)), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = _lib.lib().orp_box_iou_rotated(_lib.ptr(a), a.size(0), _lib.ptr(b), b.size(0), _lib.ptr(out),
                                            _lib.stream_of(a))
    _lib.check(rc, "orp_box_iou_rotated")
    return out
