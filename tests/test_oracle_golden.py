"""CPU: the oracle restatement (oracle/orp_oracle.c) against the committed golden vectors, which were produced by
the reference's own functions (tests/golden/make_golden.py -> oracle/_ref).  Bit-exact where the arithmetic is
+,-,*,/ only; libm-tolerant where cos/atan2/exp/log/pow are involved (same libm here, so these are exact too)."""
import os

import numpy as np


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_unit_squares_known_answer(oracle, golden_dir):
    # the one known answer the reference holds: DOTA_devkit/polyiou.cpp:131-132 -> 1/7
    g = _load(golden_dir, "polyiou_f64.npz")
    assert abs(oracle.polyiou(g["unit_p"], g["unit_q"]) - 1.0 / 7.0) < 1e-15
    assert oracle.polyiou(g["unit_p"], g["unit_q"]) == float(g["unit_iou"])
    p = g["unit_p"].astype(np.float32)[None]
    q = g["unit_q"].astype(np.float32)[None]
    assert oracle.quad_iou_matrix(p, q)[0, 0] == np.float32(0.14285713)


def test_polyiou_f64_bit_exact(oracle, golden_dir):
    g = _load(golden_dir, "polyiou_f64.npz")
    d = g["dets"]
    m = np.array([[oracle.polyiou(d[i, :8], d[j, :8]) for j in range(d.shape[0])] for i in range(d.shape[0])])
    assert np.array_equal(m, g["iou"], equal_nan=True)


def test_quad_iou_f32_bit_exact(oracle, golden_dir):
    g = _load(golden_dir, "quad_iou_nms.npz")
    for name in ("uniform", "clustered", "offset"):
        d = g["dets_" + name]
        got = oracle.quad_iou_matrix(d, d)
        assert np.array_equal(got.view(np.uint32), g["iou_" + name].view(np.uint32)), name


def test_nms_keep_sets(oracle, golden_dir):
    g = _load(golden_dir, "quad_iou_nms.npz")
    for name in ("uniform", "clustered"):
        d = g["nms_dets_" + name]
        for thr in (0.1, 0.3, 0.4):
            want = np.sort(g["nms_keep_%s_%02d_rnms" % (name, int(thr * 10))])
            assert np.array_equal(oracle.rnms(d, thr), want)
            order = oracle.sort_order(d[:, 8])
            got_poly = order[oracle.nms_sorted(d[order], thr, guard=True)]
            assert np.array_equal(got_poly, g["nms_keep_%s_%02d_poly" % (name, int(thr * 10))])
    d = g["nms_dets_dense"]
    assert np.array_equal(oracle.rnms(d, 0.4), np.sort(g["nms_keep_dense_04_rnms"]))
    # something actually gets suppressed in the clustered scene
    assert len(g["nms_keep_clustered_01_rnms"]) < 400


def test_poly_overlaps(oracle, golden_dir):
    g = _load(golden_dir, "poly_overlaps.npz")
    assert np.array_equal(oracle.poly_overlaps(g["boxes"], g["query"]), g["iou"], equal_nan=True)


def test_minarearect(oracle, golden_dir):
    g = _load(golden_dir, "minarearect.npz")
    assert np.array_equal(oracle.minarearect(g["pts"]), g["rect"])
    assert np.array_equal(oracle.minarearect(g["special"]), g["special_rect"], equal_nan=True)
    # 3x3 grid 4 wide 2 tall -> the rectangle (4,0) (0,0) (0,2) (4,2) up to fp32 rounding of cos(pi/2)
    assert np.allclose(g["special_rect"][0], [4, 0, 0, 0, 0, 2, 4, 2], atol=1e-5)


def test_convex_iou(oracle, golden_dir):
    g = _load(golden_dir, "convex_iou.npz")
    got = oracle.convex_iou(g["pts"], g["gts"])
    assert np.array_equal(got, g["iou"], equal_nan=True)
    assert (got > 0.1).sum() > 100      # the fixture exercises real overlaps, not only zeros


def test_pointwise_ops(oracle, golden_dir):
    g = _load(golden_dir, "points_justify.npz")
    assert np.array_equal(oracle.points_justify(g["demo_p"], g["demo_q"]), g["demo_out"])
    assert np.array_equal(oracle.points_justify(g["P"], g["Q"]), g["out"])
    assert 0 < g["out"].sum() < g["out"].size
    c = _load(golden_dir, "chamfer_focal.npz")
    d1, d2, i1, i2 = oracle.chamfer_forward(c["a"], c["b"])
    assert np.array_equal(d1, c["dist1"]) and np.array_equal(i1, c["idx1"])
    assert np.array_equal(d2, c["dist2"]) and np.array_equal(i2, c["idx2"])
    assert np.array_equal(oracle.focal_forward(c["logits"], c["targets"], 2.0, 0.25), c["focal_fwd"])
    assert np.array_equal(oracle.focal_backward(c["logits"], c["targets"], c["d_losses"], 2.0, 0.25), c["focal_bwd"])


def test_clip_scratch_bound(oracle):
    """The HIP kernels give each lane 8 clip slots; check the oracle never needs more on adversarial-ish inputs."""
    from orientedreppoints_amd import synthetic as S
    oracle.stats_reset()
    d = S.gen_polys(200, 5, clustered=True).astype(np.float32)
    oracle.quad_iou_matrix(d, d)
    dd, _ = S.gen_dense_scene(200, 6)
    oracle.quad_iou_matrix(dd.astype(np.float32), dd.astype(np.float32))
    # degenerate: zero-area and repeated boxes
    z = d.copy(); z[:50, 2:8] = np.tile(z[:50, 0:2], 3)
    oracle.quad_iou_matrix(z, z)
    st = oracle.stats()
    assert st["clip_overflow"] == 0 and st["max_clip_n"] <= 6, st


def test_box_iou_rotated(oracle, golden_dir):
    g = _load(golden_dir, "box_iou_rotated.npz")
    got = oracle.box_iou_rotated(g["a"], g["b"])
    assert np.array_equal(got, g["iou"], equal_nan=True)
    assert abs(float(g["unit"][0, 0]) - 0.8223) < 1e-4       # SURVEY 8c check value for 10x10 boxes offset by 0.5
    assert (got > 0.1).sum() > 20


def test_deform_conv_oracle_with_zero_offsets_is_the_convolution(oracle):
    """The checker tests/test_gpu_conv_split.py holds the tower / FPN convolutions against: the oracle's DeformConv forward
    (deform_conv_cuda_kernel.cu:84-115 samples, contracted in double) with ZERO offsets must be the ordinary convolution --
    restated here position by position in float64 numpy; odd sizes, padding 1 and 2 (dilated), batch 2."""
    rng = np.random.RandomState(3)
    for (B, C, H, W, Co, pad, dil) in ((2, 6, 5, 7, 4, 1, 1), (1, 3, 6, 4, 5, 2, 2)):
        x = rng.normal(size=(B, C, H, W)).astype(np.float32)
        w = rng.normal(size=(Co, C, 3, 3)).astype(np.float32)
        got = oracle.dcn_forward(x, np.zeros((B, 18, H, W), np.float32), w, stride=1, pad=pad, dil=dil)
        xp = np.zeros((B, C, H + 2 * pad, W + 2 * pad), np.float64)
        xp[:, :, pad:pad + H, pad:pad + W] = x
        want = np.zeros((B, Co, H, W), np.float64)
        for i in range(3):
            for j in range(3):
                patch = xp[:, :, i * dil:i * dil + H, j * dil:j * dil + W]
                want += np.einsum('bchw,oc->bohw', patch, w[:, :, i, j].astype(np.float64))
        assert got.shape == want.shape
        assert float(np.abs(got - want).max()) <= 1e-6 * max(1.0, float(np.abs(want).max()))
