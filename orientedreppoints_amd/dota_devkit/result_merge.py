"""Mirror of DOTA_devkit/ResultMerge.py (the merge step of the DOTA evaluation workflow: per-class Task1 result files
of image PATCHES -> coordinates mapped back to the original image -> polygon NMS per (class, image) -> merged files),
with the polygon NMS on the MI355X: `py_gpu_nms_poly` = `py_cpu_nms_poly` (ResultMerge.py:18-41, fp64 polyiou,
`ovr <= thresh` survives) through `orp_poly_nms_f64`.  Same function names, file formats, patch-name grammar
(`<image>__<rate>__<x>___<y>`) and `nms_thresh = 0.3` default; `mergebypoly(srcpath, dstpath)` is the entry point
(ResultMerge.py:165-172).  SURVEY 8f rank 2.
"""
import os
import re

import numpy as np
import torch

from .. import _lib

# the thresh for nms when merge image (ResultMerge.py:15)
nms_thresh = 0.3


def py_gpu_nms_poly(dets, thresh, device=None):
    """dets [N,9] float64 (8 coords + score) -> list of kept ORIGINAL indices in visiting order, identical to
    `py_cpu_nms_poly(dets, thresh)`: visiting order `scores.argsort()[::-1]` is taken on the host with numpy exactly as
    the reference does (ties included); IoU is DOTA_devkit/polyiou.cpp's fp64 arithmetic evaluated on the GPU."""
    dets = np.asarray(dets, dtype=np.float64)
    if dets.shape[0] == 0:
        return []
    order = dets[:, 8].argsort()[::-1]
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    d = torch.from_numpy(np.ascontiguousarray(dets[order])).to(dev)
    n = d.size(0)
    L = _lib.lib()
    keep = torch.empty((n,), dtype=torch.long, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = _lib.workspace(dev, L.orp_poly_nms_f64_workspace_bytes(n))
    with torch.cuda.device(dev):
        rc = L.orp_poly_nms_f64(_lib.ptr(d), n, float(thresh), _lib.ptr(keep), _lib.ptr(num), _lib.ptr(ws), ws.numel(),
                                _lib.stream_of(d))
    _lib.check(rc, "orp_poly_nms_f64")
    k = keep[:int(num.item())].cpu().numpy()
    return [int(i) for i in order[k]]


def py_cpu_nms(dets, thresh):
    """Pure numpy HBB NMS baseline (ResultMerge.py:44-74), dets [N,5] = x1,y1,x2,y2,score."""
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


def nmsbynamedict(nameboxdict, nms, thresh):
    nameboxnmsdict = {x: [] for x in nameboxdict}
    for imgname in nameboxdict:
        keep = nms(np.array(nameboxdict[imgname]), thresh)
        nameboxnmsdict[imgname] = [nameboxdict[imgname][index] for index in keep]
    return nameboxnmsdict


def poly2origpoly(poly, x, y, rate):
    origpoly = []
    for i in range(int(len(poly) / 2)):
        origpoly.append(float(poly[i * 2] + x) / float(rate))
        origpoly.append(float(poly[i * 2 + 1] + y) / float(rate))
    return origpoly


def custombasename(fullname):
    return os.path.basename(os.path.splitext(fullname)[0])


def GetFileFromThisRootDir(dir, ext=None):
    allfiles = []
    for root, _, files in os.walk(dir):
        for name in files:
            filepath = os.path.join(root, name)
            if ext is None or os.path.splitext(filepath)[1][1:] in ext:
                allfiles.append(filepath)
    return allfiles


_PAT_XY = re.compile(r'__\d+___\d+')
_PAT_RATE = re.compile(r'__([\d+\.]+)__\d+___')


def mergebase(srcpath, dstpath, nms):
    """ResultMerge.py:100-150: every result file of srcpath -> a merged file of the same name in dstpath."""
    for fullname in GetFileFromThisRootDir(srcpath):
        name = custombasename(fullname)
        dstname = os.path.join(dstpath, name + '.txt')
        with open(fullname, 'r') as f_in:
            nameboxdict = {}
            for line in f_in.readlines():
                splitline = line.strip().split(' ')
                subname = splitline[0]
                oriname = subname.split('__')[0]
                x_y = re.findall(_PAT_XY, subname)
                x_y_2 = re.findall(r'\d+', x_y[0])
                x, y = int(x_y_2[0]), int(x_y_2[1])
                rate = re.findall(_PAT_RATE, subname)[0]
                confidence = splitline[1]
                poly = list(map(float, splitline[2:]))
                det = poly2origpoly(poly, x, y, rate)
                det.append(confidence)
                det = list(map(float, det))
                nameboxdict.setdefault(oriname, []).append(det)
        nameboxnmsdict = nmsbynamedict(nameboxdict, nms, nms_thresh)
        with open(dstname, 'w') as f_out:
            for imgname in nameboxnmsdict:
                for det in nameboxnmsdict[imgname]:
                    f_out.write(imgname + ' ' + str(det[-1]) + ' ' + ' '.join(map(str, det[0:-1])) + '\n')


def mergebyrec(srcpath, dstpath):
    mergebase(srcpath, dstpath, py_cpu_nms)


def mergebypoly(srcpath, dstpath):
    """srcpath: result files before merge and nms; dstpath: result files after merge and nms (ResultMerge.py:165-172),
    with the polygon NMS on the GPU."""
    mergebase(srcpath, dstpath, py_gpu_nms_poly)
