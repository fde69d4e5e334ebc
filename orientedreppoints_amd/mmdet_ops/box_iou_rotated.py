"""Mirror of mmdet/ops/box_iou_rotated (`box_iou_rotated(boxes1[N,5], boxes2[K,5]) -> [N,K]`, box_iou_rotated.h:20-38;
theta in radians).  Like the reference's dispatcher it accepts CUDA tensors (device kernel) AND CPU tensors (the same
per-pair arithmetic compiled for the host inside liborp_hip.so: box_iou_rotated_cpu.cpp's role); mixing devices is an
error there (`AT_ASSERTM(boxes2.type().is_cuda())`) and here."""
import torch

from .. import _lib


def box_iou_rotated(boxes1, boxes2):
    if not isinstance(boxes1, torch.Tensor) or not isinstance(boxes2, torch.Tensor):
        raise TypeError("box_iou_rotated expects torch tensors")
    if boxes1.is_cuda != boxes2.is_cuda:
        raise TypeError("boxes1 and boxes2 must be on the same kind of device")
    a = boxes1.detach().float().reshape(-1, 5).contiguous()
    b = boxes2.detach().float().reshape(-1, 5).contiguous()
    out = torch.empty((a.size(0), b.size(0)), dtype=torch.float32, device=a.device)
    if not a.is_cuda:                                       # CPU branch of box_iou_rotated.h:27-32
        rc = _lib.lib().orp_box_iou_rotated_host(_lib.ptr(a), a.size(0), _lib.ptr(b), b.size(0), _lib.ptr(out))
        _lib.check(rc, "orp_box_iou_rotated_host")
        return out
    with torch.cuda.device(a.device):
        rc = _lib.lib().orp_box_iou_rotated(_lib.ptr(a), a.size(0), _lib.ptr(b), b.size(0), _lib.ptr(out),
                                            _lib.stream_of(a))
    _lib.check(rc, "orp_box_iou_rotated")
    return out
