"""Mirror of mmdet/ops/nms/nms_wrapper.py:177-199 (`rnms`) and of the extension module it calls
(mmdet/ops/nms/src/rnms_cuda.cpp:8-13 `rnms_cuda.rnms`), on the MI355X HIP library."""
import numpy as np
import torch

from .. import _lib


class _RnmsCuda(object):
    """Stands in for the pybind module `mmdet.ops.nms.rnms_cuda` (one function: rnms)."""

    @staticmethod
    def rnms(dets, threshold):
        """dets [M,9] f32 CUDA -> LongTensor of kept ORIGINAL indices, ascending (rnms_kernel.cu:261-264).
        Empty input -> empty CPU long tensor (rnms_cuda.cpp:10-11)."""
        _lib.require_cuda(dets, "dets")
        if dets.numel() == 0:
            return torch.empty((0,), dtype=torch.long, device="cpu")
        if dets.dim() != 2 or dets.size(1) != 9:
            raise ValueError("dets must be [M, 9] (8 corner coordinates + score)")
        keep, num = rnms_device(dets, threshold)
        return keep[:int(num.item())]


rnms_cuda = _RnmsCuda()


def rnms_device(dets, iou_thr, flavor=0, presorted=False, order_out=0):
    """Stream-ordered rotated NMS without any host synchronisation.

    Returns (keep int64 [M] on device, num_keep int32 [1] on device); only keep[:num_keep] is meaningful."""
    L = _lib.lib()
    d = dets.detach()
    if d.dtype != torch.float32:
        d = d.float()
    d = d.contiguous()
    n = d.size(0)
    if n > _lib.ORP_NMS_MAX_BOXES:
        raise _lib.OrpHipError("rnms: %d boxes exceeds ORP_NMS_MAX_BOXES" % n)
    keep = torch.empty((n,), dtype=torch.long, device=d.device)
    num = torch.empty((1,), dtype=torch.int32, device=d.device)
    nbytes = L.orp_rnms_workspace_bytes(n)
    ws = _lib.workspace(d.device, nbytes)
    with torch.cuda.device(d.device):
        rc = L.orp_rnms(_lib.ptr(d), n, float(iou_thr), int(flavor), int(bool(presorted)), int(order_out),
                        _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream_of(d))
    _lib.check(rc, "orp_rnms")
    return keep, num


def rnms_batched_device(dets, seg_offsets, max_seg, iou_thr, flavor=0):
    """(image x class) segments in one launch sequence.  dets [N,9]; seg_offsets int32 [S+1] on device;
    returns (keep int64 [N] -- per segment at its own offset, ascending global indices --, num_keep int32 [S])."""
    L = _lib.lib()
    d = dets.detach().float().contiguous()
    so = seg_offsets.to(device=d.device, dtype=torch.int32).contiguous()
    n, nseg = d.size(0), so.numel() - 1
    keep = torch.empty((n,), dtype=torch.long, device=d.device)
    num = torch.zeros((max(nseg, 1),), dtype=torch.int32, device=d.device)
    nbytes = L.orp_rnms_batched_workspace_bytes(n, nseg, int(max_seg))
    ws = _lib.workspace(d.device, nbytes)
    with torch.cuda.device(d.device):
        rc = L.orp_rnms_batched(_lib.ptr(d), n, _lib.ptr(so), nseg, int(max_seg), float(iou_thr), int(flavor),
                                _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(), _lib.stream_of(d))
    _lib.check(rc, "orp_rnms_batched")
    return keep, num[:nseg]


def rnms(dets, iou_thr, device_id=None):
    """Signature and behaviour of mmdet/ops/nms/nms_wrapper.py:177-199."""
    # convert dets (tensor or numpy array) to tensor
    if isinstance(dets, torch.Tensor):
        dets_th = dets
    elif isinstance(dets, np.ndarray):
        device = 'cpu' if device_id is None else 'cuda:{}'.format(device_id)
        dets_th = torch.from_numpy(dets).to(device)
    else:
        raise TypeError(
            'dets must be either a Tensor or numpy array, but got {}'.format(
                type(dets)))
    # execute cpu or cuda nms
    if dets_th.shape[0] == 0:
        inds = dets_th.new_zeros(0, dtype=torch.long)
    else:
        if dets_th.is_cuda:
            inds = rnms_cuda.rnms(dets_th, iou_thr)
        else:
            raise TypeError('dets must be cuda tensor')
    if isinstance(dets, np.ndarray):
        return dets[inds.cpu().numpy(), :], inds
    return dets[inds, :], inds


def poly_nms_gpu(dets, thresh, force_cpu=False):
    """mmdet/ops/nms/nms_wrapper.py:11-17 re-export of DOTA_devkit's poly_nms_gpu."""
    from ..dota_devkit.poly_nms_gpu import poly_nms_gpu as _impl
    return _impl(dets, thresh, force_cpu)


def soft_rnms(dets, iou_thr, method='linear', sigma=0.5, min_score=1e-3):
    """Soft rotated NMS -- CPU only, as in the reference (mmdet/ops/nms/nms_wrapper.py:120-175 over
    rnms_cpu.soft_rnms).  dets [M,9] torch tensor or numpy array; returns (new_dets [K,9], inds [K]) in the input's
    type.  The work is done by the host function `orp_soft_rnms_host` of liborp_hip.so."""
    import ctypes
    if isinstance(dets, torch.Tensor):
        is_tensor = True
        dets_np = dets.detach().cpu().numpy()
    elif isinstance(dets, np.ndarray):
        is_tensor = False
        dets_np = dets
    else:
        raise TypeError('dets must be either a Tensor or numpy array, but got {}'.format(type(dets)))
    method_codes = {'linear': 1, 'gaussian': 2, 'original': 0}
    if method not in method_codes:
        raise ValueError('Invalid method for SoftNMS: {}'.format(method))
    d = np.ascontiguousarray(dets_np, np.float32)
    m = d.shape[0]
    out = np.empty((max(m, 1), 10), np.float32)
    num = ctypes.c_int(0)
    rc = _lib.lib().orp_soft_rnms_host(d.ctypes.data_as(ctypes.c_void_p), m, float(iou_thr), method_codes[method],
                                       float(sigma), float(min_score), out.ctypes.data_as(ctypes.c_void_p),
                                       ctypes.cast(ctypes.byref(num), ctypes.c_void_p))
    _lib.check(rc, "orp_soft_rnms_host")
    res = out[:num.value]
    new_dets, inds = res[:, :9], res[:, 9]
    if is_tensor:
        return (torch.from_numpy(new_dets.copy()).to(device=dets.device, dtype=dets.dtype),
                torch.from_numpy(inds.copy()).to(device=dets.device, dtype=torch.long))
    return new_dets.astype(dets.dtype), inds.astype(np.int64)
