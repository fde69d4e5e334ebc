"""Mirror of mmdet/ops/minarearect/minarea_rect.py:4-6 and the `minarearect.minareabbox` extension function
(mmdet/ops/minarearect/src/minarearect_cuda.cpp:5-9)."""
import torch

from .. import _lib


def minareabbox(pred):
    """pred [M,18] f32 CUDA -> flat [M*8] (empty -> empty CPU float tensor, as the reference)."""
    _lib.require_cuda(pred, "pred")
    if pred.numel() == 0:
        return torch.empty((0,), dtype=torch.float32, device="cpu")
    return minaerarect_decode(pred, None, None).reshape(-1)


def minaerarect_decode(pred, centers, scales):
    """minaerarect with the optional fused decode `rect * scale + centre` (orientedreppoints_head.py:746-749).
    Stream-ordered; the result stays on the device."""
    L = _lib.lib()
    p = pred.detach()
    if p.dtype != torch.float32:
        p = p.float()
    p = p.reshape(-1, 18).contiguous()
    m = p.size(0)
    out = torch.empty((m, 8), dtype=torch.float32, device=p.device)
    c = s = None
    if centers is not None:
        c = centers.detach().float().reshape(-1, 2).contiguous()
        s = scales.detach().float().reshape(-1).contiguous()
        assert c.size(0) == m and s.numel() == m
    with torch.cuda.device(p.device):
        rc = L.orp_minarearect_decode(_lib.ptr(p), m, _lib.ptr(c), _lib.ptr(s), _lib.ptr(out), _lib.stream_of(p))
    _lib.check(rc, "orp_minarearect")
    return out


def minaerarect(pred):
    rbbox = minareabbox(pred)
    rbbox = rbbox.reshape(-1, 8)
    return rbbox
