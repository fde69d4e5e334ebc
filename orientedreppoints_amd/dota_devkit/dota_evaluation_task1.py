"""Mirror of DOTA_devkit/dota_evaluation_task1.py (the Task1 = oriented-box evaluation of the DOTA workflow: merged
per-class result files + labelTxt ground truth -> recall / precision / AP per class), with the detection-to-ground-truth
matching on the MI355X.  Same function names, file formats and defaults: `parse_gt` (:20-50), `voc_ap` (:51-84),
`voc_eval(detpath, annopath, imagesetfile, classname, ovthresh=0.5, use_07_metric=False)` (:87-239).

What moves to the GPU is the inner loop of voc_eval (:160-206): for every detection the fp64 horizontal-box pre-filter
against the ground truths of its image and `polyiou.iou_poly(GT, detection)` on the survivors, `np.max` / `np.argmax` --
one launch of `orp_voc_best_match_f64` for all detections of a class (they do not depend on the matching state).  The
tp / fp bookkeeping (:208-224, order-dependent through the `det` flags) stays on the host, as does the AP arithmetic,
so the returned arrays are the reference's bit for bit.  SURVEY 8f rank 2.
"""
import numpy as np
import torch

from .. import _lib


def parse_gt(filename):
    """labelTxt file -> list of {'name', 'difficult', 'bbox': [x1, y1, ..., x4, y4]} (lines with < 9 fields skipped)."""
    objects = []
    with open(filename, 'r') as f:
        for line in f:
            splitlines = line.strip().split(' ')
            if len(splitlines) < 9:
                continue
            object_struct = {'name': splitlines[8]}
            if len(splitlines) == 9:
                object_struct['difficult'] = 0
            elif len(splitlines) == 10:
                object_struct['difficult'] = int(splitlines[9])
            object_struct['bbox'] = [float(v) for v in splitlines[:8]]
            objects.append(object_struct)
    return objects


def voc_ap(rec, prec, use_07_metric=False):
    """VOC AP from recall / precision (11-point VOC07 metric or the area under the precision envelope)."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            p = 0 if np.sum(rec >= t) == 0 else np.max(prec[rec >= t])
            ap = ap + p / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def best_match_gpu(BB, det_img, gts, gt_off, device=None):
    """(ovmax [nd] float64, jmax [nd] int32) for detections BB [nd,8] of images det_img [nd] against the ground truths
    gts [ng,8] grouped by image through gt_off [nimg+1] -- `orp_voc_best_match_f64`."""
    nd = int(BB.shape[0])
    if nd == 0:
        return np.zeros(0), np.zeros(0, np.int32)
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    d = torch.from_numpy(np.ascontiguousarray(BB, np.float64)).to(dev)
    di = torch.from_numpy(np.ascontiguousarray(det_img, np.int32)).to(dev)
    g = torch.from_numpy(np.ascontiguousarray(gts, np.float64).reshape(-1, 8)).to(dev)
    go = torch.from_numpy(np.ascontiguousarray(gt_off, np.int32)).to(dev)
    ov = torch.empty((nd,), dtype=torch.float64, device=dev)
    jm = torch.empty((nd,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().orp_voc_best_match_f64(_lib.ptr(d), _lib.ptr(di), nd, _lib.ptr(g) if g.numel() else None,
                                               _lib.ptr(go), int(go.numel()) - 1, _lib.ptr(ov), _lib.ptr(jm),
                                               _lib.stream_of(d))
    _lib.check(rc, "orp_voc_best_match_f64")
    return ov.cpu().numpy(), jm.cpu().numpy()


def voc_eval(detpath, annopath, imagesetfile, classname, ovthresh=0.5, use_07_metric=False, best_match=None):
    """rec, prec, ap = voc_eval(detpath, annopath, imagesetfile, classname, [ovthresh], [use_07_metric]).
    detpath.format(classname): the class's result file (`image score x1 y1 ... x4 y4` per line);
    annopath.format(imagename): the image's labelTxt file; imagesetfile: one image name per line.
    `best_match` (tests): a function with `best_match_gpu`'s signature."""
    with open(imagesetfile, 'r') as f:
        imagenames = [x.strip() for x in f.readlines()]
    recs = {name: parse_gt(annopath.format(name)) for name in imagenames}

    # ground truths of this class, grouped by image
    class_recs = {}
    img_index = {}
    gt_rows, gt_off = [], [0]
    npos = 0
    for imagename in imagenames:
        R = [obj for obj in recs[imagename] if obj['name'] == classname]
        bbox = np.array([x['bbox'] for x in R])
        difficult = np.array([x['difficult'] for x in R]).astype(bool)
        npos = npos + sum(~difficult)
        if imagename not in img_index:                    # a name listed twice keeps its last record, as a dict does
            img_index[imagename] = len(img_index)
            gt_rows.append(None); gt_off.append(0)
        gt_rows[img_index[imagename]] = bbox.reshape(-1, 8) if bbox.size else np.zeros((0, 8))
        class_recs[imagename] = {'bbox': bbox, 'difficult': difficult, 'det': [False] * len(R)}
    for k, rows in enumerate(gt_rows):
        gt_off[k + 1] = gt_off[k] + len(rows)
    gts = np.concatenate(gt_rows) if gt_rows else np.zeros((0, 8))

    with open(detpath.format(classname), 'r') as f:
        splitlines = [x.strip().split(' ') for x in f.readlines()]
    image_ids = [x[0] for x in splitlines]
    confidence = np.array([float(x[1]) for x in splitlines])
    BB = np.array([[float(z) for z in x[2:]] for x in splitlines])

    sorted_ind = np.argsort(-confidence)
    BB = BB[sorted_ind, :] if len(splitlines) else np.zeros((0, 8))
    image_ids = [image_ids[x] for x in sorted_ind]
    nd = len(image_ids)
    det_img = np.array([img_index[i] for i in image_ids], np.int32)       # KeyError for an unknown image, as class_recs[...]
    ovmax_all, jmax_all = (best_match or best_match_gpu)(BB.astype(float), det_img, gts, np.array(gt_off, np.int32))

    tp = np.zeros(nd)
    fp = np.zeros(nd)
    for d in range(nd):
        R = class_recs[image_ids[d]]
        ovmax, jmax = ovmax_all[d], int(jmax_all[d])
        if ovmax > ovthresh:
            if not R['difficult'][jmax]:
                if not R['det'][jmax]:
                    tp[d] = 1.
                    R['det'][jmax] = 1
                else:
                    fp[d] = 1.
        else:
            fp[d] = 1.

    fp = np.cumsum(fp)
    tp = np.cumsum(tp)
    rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    ap = voc_ap(rec, prec, use_07_metric)
    return rec, prec, ap
