"""CPU check of the DEVICE header orp_quadfast.hpp: g++ compiles the same inline functions hipcc compiles for gfx950
(tests/host_harness/quadfast_host.cpp, -ffp-contract=off) and the register fast path of the fp32 quad IoU must be
bit-identical to the oracle (= the reference's devrIoU / devPolyIoU arithmetic) on every pair, NaNs included."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from orientedreppoints_amd import synthetic as S
from oracle import orp_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_harness", "quadfast_host.cpp")
SO = os.path.join(HERE, "host_harness", "libquadfast_host.so")


@pytest.fixture(scope="module")
def harness():
    hdrs = [os.path.join(HERE, "..", "orientedreppoints_amd", "csrc", h)
            for h in ("orp_geom.hpp", "orp_quadfast.hpp", "orp_hull.hpp")]
    if (not os.path.exists(SO)) or any(os.path.getmtime(p) > os.path.getmtime(SO) for p in [SRC] + hdrs):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    return ctypes.CDLL(SO)


def _fast(L, a, b, guard):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((len(a), len(b)), np.float32)
    st = np.zeros(2, np.int64)
    L.host_quadfast_matrix(a.ctypes.data_as(ctypes.c_void_p), len(a), b.ctypes.data_as(ctypes.c_void_p), len(b),
                           int(guard), out.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p))
    return out, (int(st[0]), int(st[1]))


def _aabb(n, lo, hi, seed):
    r = np.random.RandomState(seed)
    x0, y0 = r.randint(lo, hi, n), r.randint(lo, hi, n)
    w, h = r.randint(1, 6, n), r.randint(1, 6, n)
    return np.stack([x0, y0, x0 + w, y0, x0 + w, y0 + h, x0, y0 + h], 1).astype(np.float32)


def _angular(seed, off):
    """Boxes rotated about the ORIGIN by 1e-9 .. 1e-2 rad and moved along their ray: pairs whose fans barely overlap /
    barely separate angularly -- the margin region of the classifier's and the per-term screen's kill-3 bound."""
    r = np.random.RandomState(seed)
    base = S.gen_polys(40, seed, clustered=True)[:, :8].astype(np.float64) + off
    sets = [base]
    for d in 10.0 ** r.uniform(-9, -2, 5):
        c, s_ = np.cos(d), np.sin(d)
        p = base.reshape(-1, 4, 2)
        rot = np.stack([p[..., 0] * c - p[..., 1] * s_, p[..., 0] * s_ + p[..., 1] * c], -1).reshape(-1, 8)
        sets += [rot, rot * r.uniform(0.3, 3.0)]
    return np.concatenate(sets).astype(np.float32)


def _cases():
    rng = np.random.RandomState(0)
    d = S.gen_polys(200, 1, clustered=True)[:, :8].astype(np.float32)
    dc = S.class_offset(np.concatenate([d, np.ones((200, 1), np.float32)], 1), rng.randint(0, 15, 200))[:, :8]
    dc = dc.astype(np.float32)
    bad = d[:40].copy()
    bad[0, 0] = np.nan; bad[1, 3] = np.inf; bad[2, :] = np.nan; bad[3, 5] = -np.inf; bad[4, 2] = 3e38; bad[5, 1] = 1e20
    z = np.zeros((5, 8), np.float32)
    lines = np.array([[0, 0, 1, 1, 2, 2, 3, 3], [5, 5, 5, 5, 5, 5, 5, 5], [1, 0, 2, 0, 3, 0, 4, 0]], np.float32)
    a = _aabb(200, 0, 12, 4)
    return {
        "dense_class_offset": S.gen_dense_scene(500, 0)[0][:, :8],
        "dense_uniform": S.gen_dense_scene(500, 3, clustered=False)[0][:, :8],
        "no_class_clustered": S.gen_polys(500, 5, clustered=True)[:, :8],
        "tiny_boxes": S.gen_polys(400, 7, clustered=True, wh=(1.0, 6.0))[:, :8],
        "aabb_int_near_origin": _aabb(300, 0, 12, 1),
        "aabb_int_negative": _aabb(300, -6, 6, 2),
        "aabb_int_big_offset": _aabb(300, 0, 12, 3) + 15000,
        "aabb_clockwise": np.concatenate([a[:, [0, 1, 6, 7, 4, 5, 2, 3]], a]),
        "near_duplicates": np.concatenate([d, d + np.float32(1e-3) * rng.randn(*d.shape).astype(np.float32),
                                           np.nextafter(d, np.float32(np.inf)), d[:, [2, 3, 4, 5, 6, 7, 0, 1]]]),
        "near_dup_class_offset": np.concatenate([dc, np.nextafter(dc, np.float32(np.inf)),
                                                 np.nextafter(dc, np.float32(-np.inf)), dc[:, [6, 7, 4, 5, 2, 3, 0, 1]]]),
        "degenerate": np.concatenate([z, lines, _aabb(50, 0, 5, 5), d[:50]]),
        "centred_on_origin": S.gen_polys(300, 9)[:, :8] - 512,
        "large_containing_origin": S.gen_polys(300, 10, wh=(100, 900))[:, :8] - 512,
        "tiny_coords": S.gen_polys(200, 11)[:, :8] * 1e-6,
        "sub_eps_coords": S.gen_polys(200, 12)[:, :8] * 1e-10,
        "random_4_points": rng.uniform(0, 50, (300, 8)),
        "random_4_points_int": rng.randint(0, 8, (400, 8)),
        "nan_inf": bad,
        "angular_margins": _angular(1, 0.0),
        "angular_margins_class_offset": _angular(2, 15400.0),
    }


@pytest.mark.parametrize("guard", [False, True])
def test_register_fast_path_bit_exact(harness, guard):
    total_slow = total_far = total_terms = 0
    for name, q in _cases().items():
        q = np.ascontiguousarray(q, np.float32)
        got, (nfar, nslow) = _fast(harness, q, q, guard)
        want = O.quad_iou_matrix(q, q, guard=guard)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        if name.startswith("dense") or name.startswith("no_class"):
            total_slow += nslow - len(q)                # the diagonal (self pairs) always has zero-sign terms
            total_far += nfar
            total_terms += len(q) * (len(q) - 1)
    # on realistic scenes the classifier must resolve most pairs and the generic fallback must stay the exception
    # (that is what makes the kernel fast)
    assert total_slow / total_terms < 1e-3
    assert total_far / total_terms > 0.6


def _term(L, a, b, guard):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    out = np.empty((len(a), len(b)), np.float32)
    st = np.zeros(4, np.int64)
    L.host_quadterm_matrix(a.ctypes.data_as(ctypes.c_void_p), len(a), b.ctypes.data_as(ctypes.c_void_p), len(b),
                           int(guard), out.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p))
    return out, [int(v) for v in st]


@pytest.mark.parametrize("guard", [False, True])
def test_term_queue_composition_bit_exact(harness, guard):
    """What the NMS mask / IoU-matrix kernels run in phases A, B1, B2, B3 (orp_tile.hpp tile_drain_terms): pair
    classifier, per-TERM exact-zero screen (kill-1 / kill-3), decision tree on the surviving terms with the generic
    loop for a single term the tree gives up on, ordered sum.  Bit-identical to the oracle on every case, and the screen
    must remove about half of the terms of unresolved pairs on realistic scenes."""
    terms = unresolved = generic = 0
    for name, q in _cases().items():
        q = np.ascontiguousarray(q, np.float32)
        got, (nfar, ngen, nterm, nempty) = _term(harness, q, q, guard)
        want = O.quad_iou_matrix(q, q, guard=guard)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
        if name.startswith("dense") or name.startswith("no_class"):
            n = len(q)
            terms += nterm - 10 * n                      # self pairs: 10 sign-nonzero term pairs, all generic
            generic += ngen - 10 * n
            unresolved += n * n - nfar - nempty - n
    assert terms / unresolved < 9.5                      # 16 before the screen
    assert generic / terms < 1e-3


def test_generic_path_same_header(harness):
    q = np.ascontiguousarray(S.gen_dense_scene(300, 2)[0][:, :8], np.float32)
    out = np.empty((300, 300), np.float32)
    harness.host_quadgeneric_matrix(q.ctypes.data_as(ctypes.c_void_p), 300, q.ctypes.data_as(ctypes.c_void_p), 300, 0,
                                    out.ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(out.view(np.uint32), O.quad_iou_matrix(q, q).view(np.uint32))


def test_fp64_instantiation_matches_polyiou(harness):
    """The same header instantiated in double = the arithmetic of DOTA_devkit/polyiou.cpp (iou_poly), which the merge
    NMS kernel uses: bit-exact against the oracle's polyiou restatement (itself pinned on the reference's polyiou)."""
    rng = np.random.RandomState(0)
    d = S.gen_polys(140, 3, clustered=True)[:, :8]
    cases = {
        "clustered": d,
        "aabb_int": _aabb(80, 0, 12, 1).astype(np.float64),
        "near_dup": np.concatenate([d[:40], d[:40] + 1e-9 * rng.randn(40, 8), d[:40, [2, 3, 4, 5, 6, 7, 0, 1]]]),
        "origin": S.gen_polys(100, 9)[:, :8] - 512,
        "degenerate": np.concatenate([np.zeros((3, 8)), d[:30]]),
    }
    for name, q in cases.items():
        q = np.ascontiguousarray(q, np.float64)
        n = len(q)
        out = np.empty((n, n), np.float64)
        st = np.zeros(2, np.int64)
        harness.host_quadfast_matrix_f64(q.ctypes.data_as(ctypes.c_void_p), n, q.ctypes.data_as(ctypes.c_void_p), n,
                                         out.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p))
        want = np.array([[O.polyiou(q[i], q[j]) for j in range(n)] for i in range(n)])
        assert np.array_equal(out.view(np.uint64), want.view(np.uint64)), name
        # the term-queue composition (what the fp64 merge-NMS kernel evaluates per lane)
        out2 = np.empty((n, n), np.float64)
        st4 = np.zeros(4, np.int64)
        harness.host_quadterm_matrix_f64(q.ctypes.data_as(ctypes.c_void_p), n, q.ctypes.data_as(ctypes.c_void_p), n,
                                         out2.ctypes.data_as(ctypes.c_void_p), st4.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(out2.view(np.uint64), want.view(np.uint64)), name


def test_convex_classifier_only_claims_exact_zeros(harness):
    """convex_iou's fp64 pair classifier (hull vs gt quad), run on the CPU exactly as the kernel runs it: wherever it
    claims "every fan term is exactly 0" the oracle's IoU must be +0.0 (or NaN for a zero union) -- and it must resolve
    most pairs of a realistic layout (that is the speed-up)."""
    rng = np.random.RandomState(1)
    gts = S.gen_gts(48, 77).astype(np.float32)
    ctr = gts.reshape(-1, 4, 2).mean(1)
    sets = {
        "random": S.gen_pointsets(1500, 3),
        "on_gts": S.gen_pointsets(960, 1, around=np.repeat(ctr, 20, 0) + rng.normal(0, 8, (960, 2))),
        "same_ray": S.gen_pointsets(960, 2, around=np.repeat(ctr, 20, 0) * rng.uniform(0.3, 2.5, (960, 1))),
        "near_origin": S.gen_pointsets(400, 4, around=rng.normal(0, 10, (400, 2))),
        "negative": S.gen_pointsets(400, 5, around=rng.uniform(-600, 600, (400, 2))),
        "degenerate": np.concatenate([np.repeat(rng.uniform(0, 1000, (60, 1, 2)), 9, 1).reshape(60, 18), np.zeros((4, 18))]),
    }
    resolved = total = 0
    for name, pts in sets.items():
        pts = np.ascontiguousarray(pts, np.float32)
        for g in (gts, (gts - 512).astype(np.float32)):
            flags = np.zeros((len(pts), len(g)), np.uint8)
            harness.host_convex_far_flags(pts.ctypes.data_as(ctypes.c_void_p), len(pts), g.ctypes.data_as(ctypes.c_void_p),
                                          len(g), flags.ctypes.data_as(ctypes.c_void_p))
            want = O.convex_iou(pts, g)
            claimed = want[flags.astype(bool)]
            ok = (claimed.view(np.uint32) == 0) | np.isnan(claimed)
            assert ok.all(), (name, claimed[~ok][:5])
            if name == "random":
                resolved += int(flags.sum()); total += flags.size
    assert resolved / total > 0.7
