"""The Task1 evaluation of the DOTA workflow (dota_devkit/dota_evaluation_task1.py, mirror of
DOTA_devkit/dota_evaluation_task1.py) against values produced by the reference's own voc_eval
(tests/golden/make_golden_eval.py): recall / precision arrays and AP bit for bit."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EVAL = os.path.join(HERE, "golden", "eval")
DETPATH = os.path.join(HERE, "golden", "merge", "merged", "Task1_{:s}.txt")
ANNOPATH = os.path.join(EVAL, "labelTxt", "{:s}.txt")
IMAGESET = os.path.join(EVAL, "imageset.txt")
CLASSES = ["plane", "ship", "small-vehicle"]


def _check(best_match):
    from orientedreppoints_amd.dota_devkit import dota_evaluation_task1 as EV
    g = np.load(os.path.join(EVAL, "voc_eval.npz"))
    for cls in CLASSES:
        for thr in (0.5, 0.3):
            for m07 in (False, True):
                rec, prec, ap = EV.voc_eval(DETPATH, ANNOPATH, IMAGESET, cls, ovthresh=thr, use_07_metric=m07,
                                            best_match=best_match)
                key = "%s_%02d_%d" % (cls, int(thr * 10), int(m07))
                assert np.array_equal(rec, g["rec_" + key]), key
                assert np.array_equal(prec, g["prec_" + key]), key
                assert np.float64(ap) == g["ap_" + key], key


def test_voc_eval_host_logic_with_oracle_matcher():
    """file parsing, grouping, confidence order, tp / fp bookkeeping, AP: identical to the reference when the matcher is
    the CPU oracle (restatement of dota_evaluation_task1.py:160-206 over the polyiou port)."""
    from oracle import orp_oracle as O
    _check(lambda BB, det_img, gts, gt_off, device=None: O.voc_best_match(BB, det_img, gts, gt_off))


@pytest.mark.gpu
def test_voc_eval_gpu_matches_reference():
    import torch
    assert torch.cuda.is_available()
    _check(None)


@pytest.mark.gpu
def test_voc_best_match_kernel_vs_oracle():
    """orp_voc_best_match_f64 against the oracle: overlapping / disjoint / pre-filtered ground truths, images without
    ground truth, exact ties (first index wins), degenerate boxes (NaN IoU wins as in numpy), > 64 ground truths."""
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.dota_devkit.dota_evaluation_task1 import best_match_gpu
    from oracle import orp_oracle as O
    rng = np.random.RandomState(9)
    gts, off = [], [0]
    for k, n in enumerate((150, 0, 3, 70, 1)):
        g = S.gen_polys(n, 50 + k, clustered=True)[:, :8] if n else np.zeros((0, 8))
        gts.append(g); off.append(off[-1] + n)
    gts = np.concatenate(gts)
    gts[5] = gts[4]                                        # exact duplicate ground truth: argmax takes the first
    gts[152] = 7.0                                         # single-point ground truth (zero area)
    det_img = rng.randint(0, 5, 900).astype(np.int32)
    src = rng.randint(0, len(gts), 900)
    dets = gts[src] + rng.normal(0, 2.0, (900, 8))
    dets[::9] = S.gen_polys(100, 3)[:, :8]                 # unrelated boxes
    dets[7] = gts[4]; det_img[7] = 0                       # exact copy of the duplicated pair
    dets[11] = 7.0; det_img[11] = 2                        # degenerate detection on the degenerate ground truth
    ov, jm = best_match_gpu(dets, det_img, gts, np.array(off, np.int32))
    wov, wjm = O.voc_best_match(dets, det_img, gts, off)
    assert np.isnan(wov).any() and np.isneginf(wov).any()             # the edge cases are really in the sample
    assert np.array_equal(np.isnan(ov), np.isnan(wov))
    assert np.array_equal(ov[~np.isnan(wov)], wov[~np.isnan(wov)])     # fp64, bit for bit
    valid = ~(np.isneginf(wov))
    assert np.array_equal(jm[valid], wjm[valid]) and np.all(jm[~valid] == -1)
