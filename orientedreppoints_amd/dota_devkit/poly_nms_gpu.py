"""Mirror of DOTA_devkit/poly_nms_gpu: `poly_gpu_nms` (poly_nms.pyx:9-24), `poly_overlaps` (poly_overlaps.pyx:7-12)
and `poly_nms_gpu` (nms_wrapper.py:11-17): numpy in, numpy / list out, through the host-pointer C entry points
`_poly_nms` / `_overlaps` whose signatures are those of poly_nms.hpp:9-10 and poly_overlaps.hpp:1."""
import ctypes

import numpy as np

from .. import _lib


def poly_gpu_nms(dets, thresh, device_id=0):
    """dets ndarray[N,9] float32 -> list of kept ORIGINAL indices in score order."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    boxes_num, boxes_dim = dets.shape[0], dets.shape[1]
    if boxes_num == 0:
        return []
    keep = np.zeros(boxes_num, dtype=np.int32)
    num_out = ctypes.c_int(0)
    scores = dets[:, 8]
    order = scores.argsort()[::-1]           # the reference's exact visiting order (poly_nms.pyx:18-21)
    sorted_dets = np.ascontiguousarray(dets[order, :])
    _lib.lib()._poly_nms(keep.ctypes.data_as(ctypes.c_void_p), ctypes.cast(ctypes.byref(num_out), ctypes.c_void_p),
                         sorted_dets.ctypes.data_as(ctypes.c_void_p), boxes_num, boxes_dim, float(thresh),
                         int(device_id))
    keep = keep[:num_out.value]
    return list(order[keep])


def poly_overlaps(boxes, query_boxes, device_id=0):
    """boxes [N,5], query_boxes [K,5] float32 (cx,cy,w,h,theta) -> ndarray[N,K] float32."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32)
    query_boxes = np.ascontiguousarray(query_boxes, dtype=np.float32)
    n, k = boxes.shape[0], query_boxes.shape[0]
    overlaps = np.zeros((n, k), dtype=np.float32)
    if n == 0 or k == 0:
        return overlaps
    _lib.lib()._overlaps(overlaps.ctypes.data_as(ctypes.c_void_p), boxes.ctypes.data_as(ctypes.c_void_p),
                         query_boxes.ctypes.data_as(ctypes.c_void_p), n, k, int(device_id))
    return overlaps


def poly_nms_gpu(dets, thresh, force_cpu=False):
    """Dispatch wrapper of DOTA_devkit/poly_nms_gpu/nms_wrapper.py:11-17."""
    if dets.shape[0] == 0:
        return []
    return poly_gpu_nms(dets, thresh, device_id=0)
