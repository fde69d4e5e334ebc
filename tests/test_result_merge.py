"""The merge step of the DOTA evaluation workflow (dota_devkit/result_merge.py, mirror of DOTA_devkit/ResultMerge.py)
against files produced by the reference's own ResultMerge.mergebypoly (tests/golden/make_golden_merge.py)."""
import filecmp
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "golden", "merge", "raw")
MERGED = os.path.join(HERE, "golden", "merge", "merged")


def _same_files(a, b):
    names = sorted(os.listdir(a))
    assert names == sorted(os.listdir(b)) and names
    for n in names:
        assert filecmp.cmp(os.path.join(a, n), os.path.join(b, n), shallow=False), n


def test_merge_host_logic_with_oracle_nms(tmp_path):
    """patch-name grammar, coordinate mapping, grouping, output format: byte-identical files when the NMS is the CPU
    oracle (no GPU involved)."""
    from orientedreppoints_amd.dota_devkit import result_merge as RM
    from oracle import orp_oracle as O
    RM.mergebase(RAW, str(tmp_path), lambda dets, thr: O.py_cpu_nms_poly(dets, thr))
    _same_files(str(tmp_path), MERGED)


@pytest.mark.gpu
def test_mergebypoly_gpu_matches_reference_files(tmp_path):
    """mergebypoly with the fp64 polygon NMS on the MI355X: byte-identical to the reference's merged files."""
    import torch
    assert torch.cuda.is_available()
    from orientedreppoints_amd.dota_devkit import result_merge as RM
    RM.mergebypoly(RAW, str(tmp_path))
    _same_files(str(tmp_path), MERGED)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,thr", [(1, 0, 0.3), (64, 1, 0.3), (65, 2, 0.1), (700, 3, 0.3), (3000, 4, 0.3)])
def test_py_gpu_nms_poly_vs_oracle(n, seed, thr):
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.dota_devkit.result_merge import py_gpu_nms_poly
    from oracle import orp_oracle as O
    d = S.gen_polys(n, seed, clustered=True)
    d[:, 8] = np.round(d[:, 8], 2)                      # score ties, as in 3-decimal result files
    if n > 100:
        d[5] = d[4]; d[7, :8] = 0.0; d[9, :8] = d[9, 0]  # exact duplicate, all-zero box, single-point box (NaN IoUs)
    assert py_gpu_nms_poly(d, thr) == O.py_cpu_nms_poly(d, thr)
