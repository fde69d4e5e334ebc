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


def test_convex_giou_values_and_gradients(ref):
    # convex_giou_kernel.cu:20-804: [P,19] = 18 gradient components + GIoU
    gts = S.gen_gts(40, 61).astype(np.float32)
    ctr = np.repeat(gts.reshape(-1, 4, 2).mean(1), 10, axis=0)
    pts = S.gen_pointsets(400, 62, around=ctr).astype(np.float32)
    g = np.repeat(gts, 10, axis=0)
    a, b = ref.convex_giou(pts, g), ref.ref_convex_giou(pts, g)
    assert a.shape == b.shape == (400, 19)
    assert np.array_equal(a, b, equal_nan=True)


def test_points_justify(ref):
    # points_justify_kernel.cu:25-102 incl. the reference's own 5 x 3 demo of mmdet/ops/point_justify/test.py:10-17
    rng = np.random.RandomState(71)
    polys = S.gen_gts(30, 72).astype(np.float32)
    pts = np.concatenate([polys.reshape(-1, 2)[:60], rng.uniform(0, 1024, (500, 2)).astype(np.float32),
                          polys.reshape(-1, 4, 2).mean(1)])          # vertices (boundary hits), random, centres
    assert np.array_equal(ref.points_justify(pts, polys), ref.ref_points_justify(pts, polys))


def test_chamfer_nearest_neighbour(ref):
    # chamfer_2d.cu:12-124 NmDistanceKernel: squared distance + index of the nearest point, both directions
    rng = np.random.RandomState(81)
    a = (rng.rand(37, 40, 2) * 100).astype(np.float32)
    b = (rng.rand(37, 40, 2) * 100).astype(np.float32)
    b[:5] = a[:5]                                                     # exact ties / zero distances
    d1, d2, i1, i2 = ref.chamfer_forward(a, b)
    rd1, ri1 = ref.ref_chamfer_nn(a, b)
    rd2, ri2 = ref.ref_chamfer_nn(b, a)
    assert np.array_equal(d1, rd1) and np.array_equal(i1, ri1)
    assert np.array_equal(d2, rd2) and np.array_equal(i2, ri2)


def test_sigmoid_focal_loss(ref):
    # sigmoid_focal_loss_cuda.cu:23-97 forward / backward
    rng = np.random.RandomState(91)
    x = rng.normal(0, 3, size=(500, 15)).astype(np.float32)
    x[:4] = np.array([-90.0, -20.0, 20.0, 90.0], np.float32)[:, None]    # saturated logits
    t = rng.randint(0, 16, size=500).astype(np.int64)
    assert np.array_equal(ref.focal_forward(x, t, 2.0, 0.25), ref.ref_focal_forward(x, t, 2.0, 0.25))
    g = rng.normal(size=(500, 15)).astype(np.float32)
    assert np.array_equal(ref.focal_backward(x, t, g, 2.0, 0.25), ref.ref_focal_backward(x, t, g, 2.0, 0.25))


def test_deform_conv_im2col_and_backward(ref):
    # deform_conv_cuda_kernel.cu:190-243 (im2col), :279-436 (col2im / coordinate gradients) through the reference's own
    # device functions; DCN GEMMs follow deform_conv_cuda.cpp:262-488
    rng = np.random.RandomState(101)
    x = rng.normal(size=(2, 8, 9, 11)).astype(np.float32)
    off = rng.normal(0, 2.5, size=(2, 18, 9, 11)).astype(np.float32)       # incl. samples outside (-1, H) x (-1, W)
    w = rng.normal(0, 0.2, size=(6, 8, 3, 3)).astype(np.float32)
    a = ref.dcn_im2col(x, off, 3, 3, 1, 1, 1)
    b = ref.dcn_im2col(x, off, 3, 3, 1, 1, 1, use_ref=True)
    assert np.array_equal(a, b)
    go = rng.normal(size=(2, 6, 9, 11)).astype(np.float32)
    gi, goff, gw = ref.dcn_backward(x, off, w, go)
    ri, roff, rw = ref.dcn_backward(x, off, w, go, use_ref=True)
    # the scatter into grad_input accumulates in a different order (atomics in the reference): 1e-5 relative
    for u, v in ((gi, ri), (goff, roff), (gw, rw)):
        assert np.max(np.abs(u - v)) <= 1e-5 * max(1.0, float(np.max(np.abs(v))))


@pytest.mark.parametrize("dg", [1, 2])
def test_modulated_deform_conv_forward_and_backward(ref, dg):
    # DCNv2: deform_conv_cuda_kernel.cu:570-633 (modulated im2col), :635-693 (col2im), :695-767 (col2im_coord: offset AND
    # mask gradients) through the reference's own device functions, driven per image with batch_size = 1 like
    # deform_conv_cuda.cpp:490-685; stride 2 / dilation 2 legs cover the index arithmetic of the column layout
    rng = np.random.RandomState(131 + dg)
    for (stride, pad, dil) in ((1, 1, 1), (2, 1, 1), (1, 2, 2)):
        x = rng.normal(size=(2, 8, 9, 11)).astype(np.float32)
        Ho, Wo = ref._odim(9, pad, dil, 3, stride), ref._odim(11, pad, dil, 3, stride)
        off = rng.normal(0, 2.5, size=(2, dg * 18, Ho, Wo)).astype(np.float32)   # incl. samples outside the image
        mask = rng.uniform(0, 1, size=(2, dg * 9, Ho, Wo)).astype(np.float32)
        w = rng.normal(0, 0.2, size=(6, 8, 3, 3)).astype(np.float32)
        bias = rng.normal(size=6).astype(np.float32)
        a = ref.dcn_v2_im2col(x, off, mask, 3, 3, pad, stride, dil, dg)
        b = ref.dcn_v2_im2col(x, off, mask, 3, 3, pad, stride, dil, dg, use_ref=True)
        assert np.array_equal(a, b)                                         # same float expressions: bit for bit
        # the direct fp64-accumulated forward == the column formulation over the reference's columns
        fa = ref.dcn_forward(x, off, w, stride, pad, dil, 1, dg, mask=mask, bias=bias)
        fb = ref.dcn_v2_forward(x, off, mask, w, bias, stride, pad, dil, dg, use_ref=True)
        assert np.max(np.abs(fa - fb)) <= 1e-6 * max(1.0, float(np.max(np.abs(fb))))
        go = rng.normal(size=fb.shape).astype(np.float32)
        got = ref.dcn_v2_backward(x, off, mask, w, go, stride, pad, dil, dg)
        want = ref.dcn_v2_backward(x, off, mask, w, go, stride, pad, dil, dg, use_ref=True)
        # the reference sums over channels in float (thread-serial) and scatters with atomics; the oracle in double
        for name, u, v in zip(("input", "offset", "mask", "weight", "bias"), got, want):
            assert np.max(np.abs(u - v)) <= 1e-5 * max(1.0, float(np.max(np.abs(v)))), name
        assert float(np.max(np.abs(want[2]))) > 0.1                         # the mask gradient is really exercised
    # mask == 1 reduces to DCNv1 (the two kernel families agree)
    one = np.ones_like(mask)
    gi1, go1, gw1 = ref.dcn_backward(x, off, w, go, stride, pad, dil, dg, use_ref=True)
    gi2, go2, _, gw2, _ = ref.dcn_v2_backward(x, off, one, w, go, stride, pad, dil, dg, use_ref=True)
    for u, v in ((gi1, gi2), (go1, go2), (gw1, gw2)):
        assert np.max(np.abs(u - v)) <= 1e-5 * max(1.0, float(np.max(np.abs(v))))


def test_box_iou_rotated(ref):
    # box_iou_rotated_utils.h:314-341 single_box_iou_rotated
    a = S.gen_rboxes(60, 111).astype(np.float32)
    b = S.gen_rboxes(40, 112).astype(np.float32)
    b[:20, :2] = a[:20, :2] + 3
    u, v = ref.box_iou_rotated(a, b), ref.box_iou_rotated(a, b, use_ref=True)
    assert np.array_equal(u, v, equal_nan=True)                # bit for bit (round 6: measured on 4.3 M pairs, tightened from 1e-6)
    a = S.gen_rboxes(900, 131).astype(np.float32)
    b = S.gen_rboxes(400, 162).astype(np.float32)
    b[:300, :2] = a[:300, :2] + np.random.RandomState(0).uniform(-8, 8, (300, 2)).astype(np.float32)
    u, v = ref.box_iou_rotated(a, b), ref.box_iou_rotated(a, b, use_ref=True)
    assert np.array_equal(u, v, equal_nan=True) and (v > 0).sum() > 3000
