"""CPU: the oracle restatements of the host-side APAA logic against golden vectors produced by the REFERENCE'S OWN
PYTHON (tests/golden/make_golden_py.py executes point_assigner.py, max_iou_assigner.py and the head's methods from
/root/reference under stub parents)."""
import os

import numpy as np


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "apaa_py.npz"))


def test_point_assigner(oracle, golden_dir):
    g = _g(golden_dir)
    for pn in (1, 3):
        got = oracle.point_assign(g["points"], g["gts"], 4, pn)
        assert np.array_equal(got, g["pa_gt_inds_%d" % pn])
        assert (got > 0).sum() > 20


def test_max_iou_assigner(oracle, golden_dir):
    g = _g(golden_dir)
    ov = np.ascontiguousarray(g["overlaps"].T)
    gi, mo = oracle.max_iou_assign(ov, 0.1, 0.1, 0.0, True)
    assert np.array_equal(gi, g["mia_gt_inds"]) and np.array_equal(mo, g["mia_max_overlaps"])
    gi2, _ = oracle.max_iou_assign(ov, 0.5, 0.3, 0.2, True)
    assert np.array_equal(gi2, g["mia2_gt_inds"])
    assert set(np.unique(gi2)) >= {-1, 0}          # the don't-care band is exercised


def test_adaptive_feature_sampling_and_cosine(oracle, golden_dir):
    g = _g(golden_dir)
    for b in range(2):
        s = oracle.sample_points(g["gapf_feat"][b], 8, g["gapf_locs"][b])
        assert np.max(np.abs(s - g["gapf_out"][b].transpose(1, 2, 0))) <= 1e-5
    assert np.max(np.abs(oracle.feature_dissimilarity(g["cos_feats"]) - g["cos_out"])) <= 1e-5


def test_point_samples_selection(oracle, golden_dir):
    g = _g(golden_dir)
    pos = g["qa_pos_inds"]
    bounds = np.cumsum([0, 1024, 256, 64, 16, 4])
    lvl = np.searchsorted(bounds, pos, side="right") - 1
    keep = oracle.apaa_select(g["qa_out"], g["sel_pos_gt_inds"], lvl, int(g["sel_pos_gt_inds"].max()))
    assert keep.sum() == int(g["sel_num_pos"])
    assert np.array_equal(keep.astype(bool), g["sel_label"][pos] > 0)
