"""CPU: pin the oracle restatement against the REAL reference code (oracle/_ref/libref_orp.so = the reference's own
device functions host-compiled by oracle/build_ref.py).  Skipped where the reference tree / prebuilt .so is absent
(the golden-vector tests cover that case)."""
import numpy as np
import pytest

from orientedreppoints_amd import synthetic as S


@pytest.fixture(scope="module")
def ref(oracle):
    from oracle import build_ref
    build_ref.build()
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    return oracle


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_quad_iou_random_seeds(ref, seed):
    d = S.gen_polys(120, 100 + seed, clustered=bool(seed % 2)).astype(np.float32)
    a, b = ref.quad_iou_matrix(d, d), ref.ref_quad_iou_matrix(d, d)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    dd, _ = S.gen_dense_scene(120, 200 + seed)
    dd = dd.astype(np.float32)
    a, b = ref.quad_iou_matrix(dd, dd), ref.ref_quad_iou_matrix(dd, dd)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_rnms_cpu_core_is_the_same_arithmetic(ref):
    # mmdet/ops/nms/src/rnms_cpu.cpp rotate_iou (the reference's own CPU fp32 NMS) == rnms_kernel.cu devrIoU
    d = S.gen_polys(60, 7, clustered=True).astype(np.float32)
    for i in range(0, 60, 3):
        for j in range(60):
            a = ref.ref_pair_iou("ref_rnms_iou", d[i, :8], d[j, :8])
            b = ref.ref_pair_iou("ref_rnms_cpu_iou", d[i, :8], d[j, :8])
            assert a == b or (np.isnan(a) and np.isnan(b))


def test_nms_sweeps(ref):
    for seed, kw in ((0, {}), (1, dict(clustered=True))):
        d = S.gen_polys(400, 300 + seed, **kw).astype(np.float32)
        order = ref.sort_order(d[:, 8])
        ds = np.ascontiguousarray(d[order])
        for thr in (0.1, 0.4):
            assert np.array_equal(ref.nms_sorted(ds, thr), ref.ref_nms_sorted(ds, thr, 0))
            assert np.array_equal(ref.nms_sorted(ds, thr, guard=True), ref.ref_nms_sorted(ds, thr, 1))


def test_minarearect_and_convex(ref):
    pts = S.gen_pointsets(600, 41).astype(np.float32)
    assert np.array_equal(ref.minarearect(pts), ref.ref_minarearect(pts))
    gts = S.gen_gts(10, 42).astype(np.float32)
    ctr = np.repeat(gts.reshape(-1, 4, 2).mean(1), 30, axis=0)
    p2 = S.gen_pointsets(300, 43, around=ctr).astype(np.float32)
    assert np.array_equal(ref.convex_iou(p2, gts), ref.ref_convex_iou(p2, gts), equal_nan=True)


def test_poly_overlaps(ref):
    a = S.gen_rboxes(70, 51).astype(np.float32)
    b = S.gen_rboxes(50, 52).astype(np.float32)
    assert np.array_equal(ref.poly_overlaps(a, b), ref.ref_poly_overlaps(a, b), equal_nan=True)
