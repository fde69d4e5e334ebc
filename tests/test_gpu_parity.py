"""GPU (MI355X): the HIP path, called through the C ABI, against the CPU oracle and the committed golden vectors.

Bars (BASELINE.json north_star): bit-exact for NMS indices / IoU decisions; |delta| <= 1e-4 on IoU / loss floats.
The fp32 quad IoU core uses only + - * / and comparisons, so it is checked BIT-EXACT as well.
"""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from orientedreppoints_amd import synthetic as S  # noqa: E402


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from orientedreppoints_amd import _lib
    _lib.lib()            # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(params=[0, 6, 3], ids=["exact_fp32_mfma", "bf16_split_6", "fp16_split_3"])
def dcn_mode(request, dev):
    """The arithmetic modes of the fp32 DeformConv forward entry points: 0 = exact-fp32 MFMA (csrc/orp_dcn.hip, incl. its
    tap-granular split of multi-round launches), 6 = products of three bf16 pieces, 3 = the library default, two fp16 pieces
    per operand (csrc/orp_dcn_split.hip)."""
    from orientedreppoints_amd import _lib
    L = _lib.lib()
    assert L.orp_dcn_set_split_mode(request.param) == 0
    yield request.param
    L.orp_dcn_set_split_mode(-1)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a, dev, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev, dtype)


def _quad_iou_matrix(a, b, dev, guard=0):
    from orientedreppoints_amd import _lib
    ta, tb = _t(a, dev), _t(b, dev)
    out = torch.empty((ta.size(0), tb.size(0)), dtype=torch.float32, device=dev)
    rc = _lib.lib().orp_quad_iou_matrix(_lib.ptr(ta), ta.size(0), _lib.ptr(tb), tb.size(0), ta.size(1), guard,
                                        _lib.ptr(out), _lib.stream_of(ta))
    _lib.check(rc, "orp_quad_iou_matrix")
    return out.cpu().numpy()


# ---- rotated IoU ------------------------------------------------------------------------------------------------
def test_quad_iou_bit_exact_golden(dev, golden_dir):
    g = _g(golden_dir, "quad_iou_nms.npz")
    for name in ("uniform", "clustered", "offset"):
        d = g["dets_" + name]
        got = _quad_iou_matrix(d, d, dev)
        assert np.array_equal(got.view(np.uint32), g["iou_" + name].view(np.uint32)), name


@pytest.mark.parametrize("seed", [0, 1])
def test_quad_iou_bit_exact_oracle(dev, oracle, seed):
    d = S.gen_polys(300, 500 + seed, clustered=True).astype(np.float32)
    dd, _ = S.gen_dense_scene(300, 600 + seed)
    for x in (d, dd.astype(np.float32)):
        got = _quad_iou_matrix(x, x, dev)
        want = oracle.quad_iou_matrix(x, x)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        gotg = _quad_iou_matrix(x, x, dev, guard=1)
        wantg = oracle.quad_iou_matrix(x, x, guard=True)
        assert np.array_equal(gotg.view(np.uint32), wantg.view(np.uint32))


def test_quad_iou_degenerate(dev, oracle):
    d = S.gen_polys(64, 9).astype(np.float32)
    d[:16, 2:8] = np.tile(d[:16, 0:2], 3)          # zero-area boxes (all corners equal)
    d[16:32] = d[32:48]                             # exact duplicates
    got = _quad_iou_matrix(d, d, dev)
    want = oracle.quad_iou_matrix(d, d)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))   # NaNs included, bit for bit


# ---- rnms ---------------------------------------------------------------------------------------------------
def test_rnms_golden_keep_sets(dev, golden_dir):
    from orientedreppoints_amd.mmdet_ops import rnms
    g = _g(golden_dir, "quad_iou_nms.npz")
    for name in ("uniform", "clustered"):
        d = g["nms_dets_" + name]
        for thr in (0.1, 0.3, 0.4):
            dets, inds = rnms(_t(d, dev), thr)
            want = np.sort(g["nms_keep_%s_%02d_rnms" % (name, int(thr * 10))])
            assert inds.dtype == torch.long and inds.is_cuda
            assert np.array_equal(inds.cpu().numpy(), want), (name, thr)
            assert np.array_equal(dets.cpu().numpy(), d[want])
    d = g["nms_dets_dense"]
    _, inds = rnms(_t(d, dev), 0.4)
    assert np.array_equal(inds.cpu().numpy(), np.sort(g["nms_keep_dense_04_rnms"]))


@pytest.mark.parametrize("n,seed,clustered", [(1, 0, False), (63, 1, True), (64, 2, True), (65, 3, True),
                                              (500, 0, False), (1000, 4, True), (2000, 5, True)])
def test_rnms_vs_oracle(dev, oracle, n, seed, clustered):
    from orientedreppoints_amd.mmdet_ops import rnms
    d = S.gen_polys(n, seed, clustered=clustered).astype(np.float32)
    for thr in (0.1, 0.4):
        _, inds = rnms(_t(d, dev), thr)
        assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, thr)), (n, thr)


def test_rnms_class_offset_dense_scene(dev, oracle):
    """config-4 stress: 2000 dets, 15 classes folded in by the class-offset trick -- fp32 cancellation included."""
    from orientedreppoints_amd.mmdet_ops import rnms
    d, _ = S.gen_dense_scene(2000, 31)
    d = d.astype(np.float32)
    _, inds = rnms(_t(d, dev), 0.4)
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.4))


def test_rnms_full_size_and_properties(dev, oracle):
    """BASELINE's maximum candidate count per image (2000 + 2000 + 1024 + 256 + 64 = 5344 class-offset boxes: 64 x 64
    tiles, several term-queue chunks per tile) against the oracle, then size-independent properties at 12 000 boxes:
    keep is ascending and unique, re-running NMS on the kept boxes keeps every one of them (idempotence), and a kept
    box never overlaps an earlier (higher-score) kept box by more than the threshold."""
    from orientedreppoints_amd.mmdet_ops import rnms
    d, _ = S.gen_dense_scene(5344, 1)
    d = d.astype(np.float32)
    _, inds = rnms(_t(d, dev), 0.4)
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.4))
    d, _ = S.gen_dense_scene(12000, 7)
    d = d.astype(np.float32)
    kept, inds = rnms(_t(d, dev), 0.4)
    k = inds.cpu().numpy()
    assert np.all(np.diff(k) > 0) and k.min() >= 0 and k.max() < len(d)
    kept2, inds2 = rnms(kept, 0.4)
    assert inds2.numel() == kept.size(0)
    kk = kept.cpu().numpy()
    order = np.argsort(-kk[:, 8], kind="stable")[:600]                 # spot-check the 600 best survivors pairwise
    iou = oracle.quad_iou_matrix(kk[order, :8], kk[order, :8])
    iu = np.triu_indices(len(order), 1)
    assert not np.any(iou[iu] > 0.4)


def test_rnms_sparse_and_dense_sweep_paths(dev, oracle):
    """The sweep runs out of LDS when the mask has <= 8192 non-zero words and falls back to the dense block-row pass
    otherwise: a heavily overlapping cluster (tens of thousands of hits) must take the fallback and still reproduce
    the oracle's keep set; a sparse scene of the same size takes the LDS path."""
    from orientedreppoints_amd.mmdet_ops import rnms
    rng = np.random.RandomState(8)
    n = 1800
    # dense: all boxes within a few pixels of 6 hubs -> almost every pair overlaps
    hubs = rng.uniform(200, 800, (6, 2))
    ctr = hubs[rng.randint(0, 6, n)] + rng.normal(0, 3.0, (n, 2))
    w, h, th = rng.uniform(30, 60, n), rng.uniform(30, 60, n), rng.uniform(-0.3, 0.3, n)
    c, s_ = np.cos(th), np.sin(th)
    dx = np.stack([-w / 2, w / 2, w / 2, -w / 2], 1); dy = np.stack([-h / 2, -h / 2, h / 2, h / 2], 1)
    xs = ctr[:, :1] + c[:, None] * dx - s_[:, None] * dy
    ys = ctr[:, 1:] + s_[:, None] * dx + c[:, None] * dy
    dense = np.concatenate([np.stack([xs, ys], 2).reshape(n, 8), rng.uniform(0.05, 1, (n, 1))], 1).astype(np.float32)
    hits = (oracle.quad_iou_matrix(dense[:, :8], dense[:, :8]) > 0.4).sum()
    assert hits > 200000                                   # far more non-zero mask words than the LDS list holds
    for d in (dense, S.gen_polys(n, 4, clustered=False).astype(np.float32)):
        _, inds = rnms(_t(d, dev), 0.4)
        assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.4))


def test_rnms_ties_and_api_edges(dev, oracle):
    from orientedreppoints_amd.mmdet_ops import rnms, rnms_cuda
    d = S.gen_polys(300, 8, clustered=True).astype(np.float32)
    d[:, 8] = np.round(d[:, 8] * 4) / 4                 # heavy score ties -> (score desc, index asc) order matters
    _, inds = rnms(_t(d, dev), 0.3)
    assert np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.3))
    # empty input -> empty long tensor; CPU tensor -> TypeError  (nms_wrapper.py:190-197)
    e = torch.zeros((0, 9), device=dev)
    dets, inds = rnms(e, 0.4)
    assert inds.numel() == 0 and inds.dtype == torch.long and dets.shape == (0, 9)
    with pytest.raises(TypeError):
        rnms(torch.zeros((3, 9)), 0.4)
    with pytest.raises(TypeError):
        rnms([1, 2, 3], 0.4)
    assert rnms_cuda.rnms(e, 0.4).device.type == "cpu"
    # numpy input with device_id
    dn, inds = rnms(d, 0.3, device_id=0)
    assert isinstance(dn, np.ndarray) and np.array_equal(inds.cpu().numpy(), oracle.rnms(d, 0.3))


def test_rnms_batched_segments(dev, oracle):
    from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_batched_device
    sizes = [0, 1, 130, 64, 500, 0, 333]
    parts = [S.gen_polys(n, 70 + i, clustered=True).astype(np.float32) for i, n in enumerate(sizes)]
    d = np.concatenate(parts, 0)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    keep, num = rnms_batched_device(_t(d, dev), torch.from_numpy(off), max(sizes), 0.3)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    for s, n in enumerate(sizes):
        want = oracle.rnms(parts[s], 0.3) + off[s]
        assert num[s] == len(want)
        assert np.array_equal(keep[off[s]:off[s] + num[s]], want), s


# ---- DOTA_devkit host-pointer API -----------------------------------------------------------------------------
def test_poly_gpu_nms_and_overlaps_host_api(dev, oracle, golden_dir):
    from orientedreppoints_amd.dota_devkit.poly_nms_gpu import poly_gpu_nms, poly_overlaps, poly_nms_gpu
    d = S.gen_polys(500, 0).astype(np.float32)            # config 0 of BASELINE.json
    for thr in (0.3, 0.1, 0.4):
        got = poly_gpu_nms(d, thr)
        assert [int(x) for x in got] == oracle.poly_gpu_nms(d, thr)
    assert poly_nms_gpu(np.zeros((0, 9), np.float32), 0.3) == []
    g = _g(golden_dir, "poly_overlaps.npz")
    got = poly_overlaps(g["boxes"], g["query"])
    assert got.shape == g["iou"].shape
    assert np.nanmax(np.abs(got - g["iou"])) <= 1e-4
    # round 6: RotBox2Poly's cos / sin are the host C library's (csrc/orp_libm.hpp) -> the golden's bits (reference compiled for the host)
    assert np.array_equal(got, g["iou"], equal_nan=True)


def test_poly_nms_matches_fp64_cpu_reference_on_config0(dev, oracle):
    """fp32 GPU poly NMS vs the fp64 CPU path (polyiou + py_cpu_nms_poly) on config 0: same keep set."""
    from orientedreppoints_amd.dota_devkit.poly_nms_gpu import poly_gpu_nms
    d64 = S.gen_polys(500, 0)
    for thr in (0.3, 0.1, 0.4):
        assert sorted(int(x) for x in poly_gpu_nms(d64.astype(np.float32), thr)) == sorted(oracle.py_cpu_nms_poly(d64, thr))


# ---- minaerarect ------------------------------------------------------------------------------------------------
def test_minarearect(dev, oracle, golden_dir):
    """a4, BIT-EXACT since round 6: the kernel's cosines are the host C library's cosf (csrc/orp_libm.hpp), which is what the
    reference compiled for the host -- the oracle -- evaluates; rounds 1-5 used a correctly rounded cosine and differed from it
    on rounding-level ties between two candidate edge directions (the 1e-4 bar and the a15 "tie rule" of those rounds)."""
    from orientedreppoints_amd.mmdet_ops import minaerarect
    g = _g(golden_dir, "minarearect.npz")
    got = minaerarect(_t(g["pts"], dev)).cpu().numpy()
    assert got.shape == g["rect"].shape
    assert np.array_equal(got, g["rect"]), "golden (reference compiled for the host): expected the same bits"
    sp = minaerarect(_t(g["special"], dev)).cpu().numpy()
    assert np.array_equal(sp, g["special_rect"], equal_nan=True)
    pts = S.gen_pointsets(5344, 77).astype(np.float32)       # max candidates per image at test time
    got = minaerarect(_t(pts, dev)).cpu().numpy()
    want = oracle.minarearect(pts)
    assert np.array_equal(got, want)
    # empty -> empty CPU tensor reshaped [0,8]; fused decode = rect*scale + centre
    assert minaerarect(torch.zeros((0, 18), device=dev)).shape == (0, 8)
    from orientedreppoints_amd.mmdet_ops.minarea_rect import minaerarect_decode
    c = np.random.RandomState(0).uniform(0, 1024, (pts.shape[0], 2)).astype(np.float32)
    s = np.random.RandomState(1).choice([8, 16, 32, 64, 128], pts.shape[0]).astype(np.float32)
    dec = minaerarect_decode(_t(pts, dev), _t(c, dev), _t(s, dev)).cpu().numpy()
    assert np.array_equal(dec, got * s[:, None] + np.tile(c, 4))         # one multiply, one add per coordinate, unfused


def _minarearect_families(n, seed=0):
    """Point-set families for the bitwise sweep: what a trained head emits (jittered rotated grids), what the INITIAL stage emits
    (exact regular 3x3 grids, axis aligned / rotated / on an integer lattice: every one a tie between two edge directions by
    construction) and degenerate hulls."""
    rng = np.random.RandomState(seed)
    g = np.array([[x, y] for y in (-1.0, 0.0, 1.0) for x in (-1.0, 0.0, 1.0)])
    fam = {"random": S.gen_pointsets(n, seed + 1)}
    sx, sy = rng.uniform(0.5, 40, (n, 1)), rng.uniform(0.5, 40, (n, 1))
    c = rng.uniform(0, 1024, (n, 1, 2))
    fam["grid_axis"] = (np.stack([g[None, :, 0] * sx, g[None, :, 1] * sy], 2) + c).reshape(n, 18)
    th = rng.uniform(-np.pi, np.pi, (n, 1))
    px, py = g[None, :, 0] * sx, g[None, :, 1] * sy
    fam["grid_rot"] = (np.stack([np.cos(th) * px - np.sin(th) * py, np.sin(th) * px + np.cos(th) * py], 2) + c).reshape(n, 18)
    k = rng.randint(1, 30, (n, 1)).astype(np.float64)
    ci = rng.randint(0, 1024, (n, 1, 2)).astype(np.float64)
    a, b = rng.randint(-6, 7, (n, 1)).astype(np.float64), rng.randint(-6, 7, (n, 1)).astype(np.float64)
    a[(a == 0) & (b == 0)] = 1
    fam["grid_int"] = (np.stack([g[None, :, 0] * a * k - g[None, :, 1] * b * k,
                                 g[None, :, 0] * b * k + g[None, :, 1] * a * k], 2) + ci).reshape(n, 18)
    m = max(n // 8, 1)
    t = rng.uniform(-30, 30, (m, 9, 1))
    fam["collinear"] = (rng.uniform(0, 1024, (m, 1, 2)) + t * rng.normal(size=(m, 1, 2))).reshape(m, 18)
    dup = S.gen_pointsets(m, seed + 2).reshape(m, 9, 2)
    dup[:, 3:] = dup[:, rng.randint(0, 3, 6)]
    fam["duplicate"] = dup.reshape(m, 18)
    fam["tiny"] = (rng.uniform(0, 1024, (m, 1, 2)) + rng.normal(0, 1e-3, (m, 9, 2))).reshape(m, 18)
    return {k_: v.astype(np.float32) for k_, v in fam.items()}


def test_minarearect_one_million_sets_bitwise(dev, oracle):
    """Round-5 verdict, next 2a: count the point sets on which ANY of the 8 output floats differs in ANY bit from the oracle
    (= the reference's minBoundingRect / Jarvis compiled for the host), ~1 M sets, exact regular grids included.  Must be 0."""
    from orientedreppoints_amd.mmdet_ops import minaerarect
    import conftest
    total = bad = 0
    for name, pts in _minarearect_families(230000, seed=6).items():
        got = minaerarect(_t(pts, dev)).cpu().numpy()
        want = oracle.minarearect(pts)
        neq = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        rows = int(np.count_nonzero(neq.any(1)))
        total += len(pts); bad += rows
        assert rows == 0, "%s: %d of %d point sets differ from the oracle bitwise" % (name, rows, len(pts))
    conftest.REPORT.append("minaerarect vs oracle, bitwise: %d of %d point sets differ (7 families incl. exact regular grids)" % (bad, total))
    assert total >= 1000000


def test_device_exp_log_pow_are_the_host_librarys(dev):
    """csrc/orp_libm.hpp on gfx950 against the C library of THIS host: expf / logf on 4 M floats drawn over the whole range (bit
    patterns), powf on 4 M (x, y) pairs incl. the focal loss's (p in [0, 1], gamma) region.  Bitwise (NaN == NaN)."""
    import ctypes
    from orientedreppoints_amd import _lib
    L = _lib.lib()
    harness = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness", "liblibm_host.so")
    if not os.path.exists(harness):
        import subprocess
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", harness,
                               os.path.join(os.path.dirname(harness), "libm_host.cpp")])
    libm = ctypes.CDLL("libm.so.6")
    rng = np.random.RandomState(3)
    bits = rng.randint(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    xs = np.concatenate([x[:2_000_000], rng.uniform(-100, 100, 2_000_000).astype(np.float32)])
    for which, fn in ((2, np.exp), (3, np.log)):
        xd = _t(xs, dev)
        out = torch.empty_like(xd)
        assert L.orp_libm_eval(_lib.ptr(xd), None, xs.size, which, _lib.ptr(out), _lib.stream_of(xd)) == 0
        H = ctypes.CDLL(harness)                                   # the g++ build of the same header == libm (tests/test_libm_host.py)
        want = np.empty_like(xs)
        H.host_libm_eval(xs.ctypes.data_as(ctypes.c_void_p), xs.size, which, want.ctypes.data_as(ctypes.c_void_p))
        got = out.cpu().numpy()
        assert np.array_equal(got.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)]) and np.all(np.isnan(got[np.isnan(want)]))
        # and directly against libm on a sample
        f = getattr(libm, "expf" if which == 2 else "logf"); f.restype = ctypes.c_float; f.argtypes = [ctypes.c_float]
        idx = rng.randint(0, xs.size, 20000)
        ref = np.array([f(float(v)) for v in xs[idx]], np.float32)
        assert np.array_equal(got[idx].view(np.uint32)[~np.isnan(ref)], ref.view(np.uint32)[~np.isnan(ref)])
    px = np.concatenate([rng.uniform(0, 1, 2_000_000), np.abs(x[:2_000_000])]).astype(np.float32)
    py = np.concatenate([rng.choice([2.0, 1.5, 0.5, 3.0], 2_000_000), x[2_000_000:]]).astype(np.float32)
    out = torch.empty(px.size, device=dev)
    xd, yd = _t(px, dev), _t(py, dev)
    assert L.orp_libm_eval(_lib.ptr(xd), _lib.ptr(yd), px.size, 4, _lib.ptr(out), _lib.stream_of(xd)) == 0
    libm.powf.restype = ctypes.c_float; libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
    idx = rng.randint(0, px.size, 40000)
    ref = np.array([libm.powf(float(a), float(b)) for a, b in zip(px[idx], py[idx])], np.float32)
    got = out.cpu().numpy()[idx]
    ok = ~np.isnan(ref)
    assert np.array_equal(got.view(np.uint32)[ok], ref.view(np.uint32)[ok]) and np.all(np.isnan(got[~ok]))


def test_device_cos_sin_are_the_host_librarys(dev):
    """csrc/orp_libm.hpp on gfx950 against the C library of THIS host (glibc): every float in (-4, 4), both functions, bitwise.
    The g++ build of the same header is checked against libm over (-96, 96) by tests/test_libm_host.py."""
    import ctypes
    from orientedreppoints_amd import _lib
    L = _lib.lib()
    libm = ctypes.CDLL("libm.so.6")
    hi = int(np.float32(4.0).view(np.uint32))
    sample = np.random.RandomState(0).randint(0, hi, 20000).astype(np.uint32)
    for which, fn in ((0, libm.cosf), (1, libm.sinf)):
        fn.restype = ctypes.c_float; fn.argtypes = [ctypes.c_float]
        # (a) every float of (-4, 4): the device result equals the float-rounded double function EXCEPT where the C library's
        #     does not either -- checked exhaustively against the g++ build in the CPU suite; here: exhaustive self-consistency
        #     of the two signs, and a 40 000-point direct comparison with libm itself
        for neg in (0, 1):
            bits = sample | np.uint32(0x80000000 if neg else 0)
            x = bits.view(np.float32)
            xd = _t(x, dev)
            out = torch.empty_like(xd)
            assert L.orp_libm_eval(_lib.ptr(xd), None, x.size, which, _lib.ptr(out), _lib.stream_of(xd)) == 0
            want = np.array([fn(float(v)) for v in x], np.float32)
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        allbits = np.arange(0, hi, dtype=np.uint32)
        xd = _t(allbits.view(np.float32), dev)
        pos, negv = torch.empty_like(xd), torch.empty_like(xd)
        assert L.orp_libm_eval(_lib.ptr(xd), None, allbits.size, which, _lib.ptr(pos), _lib.stream_of(xd)) == 0
        xn = -xd
        assert L.orp_libm_eval(_lib.ptr(xn), None, allbits.size, which, _lib.ptr(negv), _lib.stream_of(xd)) == 0
        assert torch.equal(pos, -negv if which else negv)             # cos even, sin odd: exact symmetries of the algorithm
        dbl = (np.sin if which else np.cos)(allbits.view(np.float32).astype(np.float64))
        ulp = np.abs(pos.cpu().numpy().astype(np.float64) - dbl) / np.maximum(np.spacing(np.abs(dbl).astype(np.float32)), 1e-45)
        assert float(ulp.max()) <= 0.57                                # the algorithm's documented worst case (0.56 ulp)


# ---- convex_iou ---------------------------------------------------------------------------------------------------
def test_convex_iou(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_ops import convex_iou, convex_overlaps
    g = _g(golden_dir, "convex_iou.npz")
    got = convex_iou(_t(g["pts"], dev), _t(g["gts"], dev)).cpu().numpy()
    assert got.shape == g["iou"].shape
    assert np.array_equal(np.isnan(got), np.isnan(g["iou"]))
    assert np.nanmax(np.abs(got - g["iou"])) <= 1e-4
    assert np.array_equal(got, g["iou"], equal_nan=True), "fp64 + - * / only: expected bit-exact"
    ov = convex_overlaps(_t(g["gts"], dev), _t(g["pts"], dev)).cpu().numpy()
    assert np.array_equal(ov, got.T, equal_nan=True)
    # k = 1 and k = 256 against the oracle at N = one 1024^2 image's coarse levels
    for k, seed in ((1, 1), (37, 2)):
        gts = S.gen_gts(k, 900 + seed).astype(np.float32)
        pts = S.gen_pointsets(1364, 901 + seed).astype(np.float32)
        got = convex_iou(_t(pts, dev), _t(gts, dev)).cpu().numpy()
        assert np.array_equal(got, oracle.convex_iou(pts, gts), equal_nan=True)


def test_convex_iou_far_pair_classifier_cases(dev, oracle):
    """The exact-zero classifier in front of the fp64 fan must never change a bit: point sets on top of, next to, far
    from and angularly aligned with the gts; sets at / around the coordinate origin; degenerate sets; negative
    coordinates; the full first FPN level of a 1024^2 image."""
    from orientedreppoints_amd.mmdet_ops import convex_iou
    rng = np.random.RandomState(5)
    gts = S.gen_gts(24, 77).astype(np.float32)
    ctr = gts.reshape(-1, 4, 2).mean(1)
    cases = {
        "on_gts": S.gen_pointsets(600, 1, around=np.repeat(ctr, 25, 0) + rng.normal(0, 6, (600, 2))),
        "same_ray": S.gen_pointsets(480, 2, around=np.repeat(ctr, 20, 0) * rng.uniform(0.3, 2.5, (480, 1))),
        "random": S.gen_pointsets(800, 3),
        "near_origin": S.gen_pointsets(300, 4, around=rng.normal(0, 10, (300, 2))),
        "negative": S.gen_pointsets(300, 5, around=rng.uniform(-600, 600, (300, 2))),
        "degenerate": np.concatenate([np.repeat(rng.uniform(0, 1000, (50, 1, 2)), 9, 1).reshape(50, 18),
                                      np.zeros((5, 18)),
                                      np.stack([np.linspace(0, 8, 9)] * 2, 1).reshape(1, 18) + rng.uniform(0, 900, (40, 1))]),
    }
    for name, pts in cases.items():
        pts = np.ascontiguousarray(pts, np.float32)
        for g in (gts, (gts - 512).astype(np.float32), gts[:1] * 0):
            got = convex_iou(_t(pts, dev), _t(g, dev)).cpu().numpy()
            want = oracle.convex_iou(pts, g)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name
    # level 0 of a 1024^2 image (128 x 128 locations) against 12 gts
    yy, xx = np.meshgrid(np.arange(128), np.arange(128), indexing="ij")
    around = np.stack([xx.reshape(-1) * 8.0 + 4, yy.reshape(-1) * 8.0 + 4], 1)
    pts = np.ascontiguousarray(S.gen_pointsets(128 * 128, 6, around=around), np.float32)
    got = convex_iou(_t(pts, dev), _t(gts[:12], dev)).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), oracle.convex_iou(pts, gts[:12]).view(np.uint32))


# ---- pointwise ops -----------------------------------------------------------------------------------------------
def test_points_justify(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_ops import pointsJf, points_in_quad_aligned
    g = _g(golden_dir, "points_justify.npz")
    for p, q, want in ((g["demo_p"], g["demo_q"], g["demo_out"]), (g["P"], g["Q"], g["out"])):
        out = torch.full((p.shape[0], q.shape[0]), -1.0, device=dev)
        assert pointsJf(_t(p, dev), _t(q, dev), out) == 1
        assert np.array_equal(out.cpu().numpy(), want)
    quads = S.gen_gts(700, 5).astype(np.float32)
    pts = S.gen_pointsets(700, 6, around=quads.reshape(-1, 4, 2).mean(1)).astype(np.float32)
    got = points_in_quad_aligned(_t(pts, dev), _t(quads, dev)).cpu().numpy()
    assert np.array_equal(got, oracle.points_in_quad_aligned(pts, quads))
    assert 0 < got.sum() < got.size


def test_chamfer_and_focal(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_ops import ChamferDistance2D, Chamfer2D, sigmoid_focal_loss
    c = _g(golden_dir, "chamfer_focal.npz")
    a, b = _t(c["a"], dev), _t(c["b"], dev)
    d1, d2, i1, i2 = Chamfer2D()(a, b)
    assert np.array_equal(i1.cpu().numpy(), c["idx1"]) and np.array_equal(i2.cpu().numpy(), c["idx2"])
    assert np.array_equal(d1.cpu().numpy(), c["dist1"]) and np.array_equal(d2.cpu().numpy(), c["dist2"])
    cd = ChamferDistance2D(a, b).cpu().numpy()
    want = 0.05 * (np.sqrt(np.maximum(c["dist1"], 1e-12)).mean(-1) + np.sqrt(np.maximum(c["dist2"], 1e-12)).mean(-1)) / 2
    assert np.allclose(cd, want, rtol=1e-5, atol=1e-6)
    # backward of the chamfer op against the oracle's scatter
    a2 = a.clone().requires_grad_(True); b2 = b.clone().requires_grad_(True)
    o1, o2, _, _ = Chamfer2D()(a2, b2)
    w1 = torch.linspace(0.5, 1.5, o1.numel(), device=dev).reshape(o1.shape)
    w2 = torch.linspace(1.5, 0.5, o2.numel(), device=dev).reshape(o2.shape)
    ((o1 * w1).sum() + (o2 * w2).sum()).backward()
    ga, gb = oracle.chamfer_backward(c["a"], c["b"], w1.cpu().numpy(), w2.cpu().numpy(), c["idx1"], c["idx2"])
    assert np.allclose(a2.grad.cpu().numpy(), ga, rtol=1e-4, atol=1e-4)
    assert np.allclose(b2.grad.cpu().numpy(), gb, rtol=1e-4, atol=1e-4)
    # focal: BIT-EXACT since round 6 (expf / logf / powf are the host C library's algorithms, csrc/orp_libm.hpp; rounds 1-5: 1e-4,
    # "exp/log/pow come from different libms"); the golden is the reference's kernel compiled for the host
    x = _t(c["logits"], dev).requires_grad_(True)
    t = torch.from_numpy(c["targets"]).to(dev)
    loss = sigmoid_focal_loss(x, t, 2.0, 0.25)
    assert np.array_equal(loss.detach().cpu().numpy(), c["focal_fwd"])
    loss.backward(_t(c["d_losses"], dev))
    assert np.array_equal(x.grad.cpu().numpy(), c["focal_bwd"])
    # ... and against the oracle on 200 000 logits over the whole range the head produces (incl. saturated ones), three gammas
    rng = np.random.RandomState(5)
    xl = rng.permutation(np.concatenate([rng.normal(0, 4, 150000), rng.uniform(-90, 90, 45000)])).astype(np.float32).reshape(-1, 15)
    tl = rng.randint(0, 16, xl.shape[0]).astype(np.int64)
    dl = rng.uniform(0.2, 2.0, xl.shape).astype(np.float32)
    from orientedreppoints_amd.mmdet_ops.sigmoid_focal_loss import sigmoid_focal_loss_cuda as ext
    for gamma, alpha in ((2.0, 0.25), (1.5, 0.5), (0.5, 0.75)):
        got = ext.forward(_t(xl, dev), torch.from_numpy(tl).to(dev), 15, gamma, alpha).cpu().numpy()
        assert np.array_equal(got, oracle.focal_forward(xl, tl, gamma, alpha), equal_nan=True)
        gb_ = ext.backward(_t(xl, dev), torch.from_numpy(tl).to(dev), _t(dl, dev), 15, gamma, alpha).cpu().numpy()
        assert np.array_equal(gb_, oracle.focal_backward(xl, tl, dl, gamma, alpha), equal_nan=True)
    with pytest.raises(RuntimeError):
        sigmoid_focal_loss(torch.zeros(2, 15), torch.zeros(2, dtype=torch.long), 2.0, 0.25)


# ---- deformable convolution -----------------------------------------------------------------------------------------
def _dcn_case(seed, B, C, H, W, Cout, std_off=2.0):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(B, C, H, W)).astype(np.float32)
    off = rng.normal(0, std_off, size=(B, 18, H, W)).astype(np.float32)
    w = rng.normal(0, 0.05, size=(Cout, C, 3, 3)).astype(np.float32)
    return x, off, w


def _rel_err(got, want):
    return np.max(np.abs(got - want)) / max(1e-6, np.max(np.abs(want)))


@pytest.mark.parametrize("B,C,H,W,Cout", [(1, 64, 9, 11, 64), (2, 256, 16, 16, 256), (1, 32, 5, 40, 128),
                                          (1, 96, 7, 7, 192)])
def test_dcn_forward_mfma_vs_oracle(dev, oracle, B, C, H, W, Cout, dcn_mode):
    """MFMA implicit GEMM (fp32-exact MFMA) against the oracle's double-accumulated contraction: <= 1e-4 relative."""
    from orientedreppoints_amd.mmdet_ops import deform_conv
    x, off, w = _dcn_case(0, B, C, H, W, Cout)
    want = oracle.dcn_forward(x, off, w, stride=1, pad=1, dil=1)
    got = deform_conv(_t(x, dev), _t(off, dev), _t(w, dev), 1, 1, 1, 1, 1, 64)
    assert got.shape == want.shape and got.is_contiguous()
    assert _rel_err(got.cpu().numpy(), want) <= 1e-4
    # channels-last in -> channels-last out, same numbers
    xcl = _t(x, dev).contiguous(memory_format=torch.channels_last)
    got2 = deform_conv(xcl, _t(off, dev), _t(w, dev), 1, 1, 1, 1, 1, 64)
    assert got2.is_contiguous(memory_format=torch.channels_last)
    assert _rel_err(got2.cpu().numpy(), want) <= 1e-4


def test_dcn_forward_multi_level_and_offsets_out_of_range(dev, oracle, dcn_mode):
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
    cases = [_dcn_case(10 + i, 1, 64, h, h, 64, std_off=4.0) for i, h in enumerate((16, 8, 4, 2, 1))]
    w = cases[0][2]
    outs = deform_conv_forward_multi([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev), 1, 1, 1)
    for (x, off, _), o in zip(cases, outs):
        assert _rel_err(o.cpu().numpy(), oracle.dcn_forward(x, off, w)) <= 1e-4


def test_dcn_pair_launch_equals_two_launches_and_oracle(dev, oracle, dcn_mode):
    """orp_dcn_forward_pair: the head's two DeformConvs (same offsets) in ONE launch -- against the oracle on a small
    multi-level case, and bit-identical to two single launches of the same kernel generation at the head's channel
    count (the per-layer accumulation order is the same)."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, deform_conv_forward_pair
    rng = np.random.RandomState(11)
    shapes = [(10, 12), (5, 7), (3, 3)]
    xa = [rng.normal(size=(2, 128, h, w)).astype(np.float32) for h, w in shapes]
    xb = [rng.normal(size=(2, 128, h, w)).astype(np.float32) for h, w in shapes]
    off = [rng.normal(0, 2.5, size=(2, 18, h, w)).astype(np.float32) for h, w in shapes]      # incl. out-of-range samples
    wa = rng.normal(0, 0.05, size=(64, 128, 3, 3)).astype(np.float32)
    wb = rng.normal(0, 0.05, size=(64, 128, 3, 3)).astype(np.float32)
    oa, ob = deform_conv_forward_pair([_t(a, dev) for a in xa], [_t(b, dev) for b in xb], [_t(o, dev) for o in off],
                                      _t(wa, dev), _t(wb, dev), 1, 1, 1, relu=False)
    for i in range(len(shapes)):
        assert _rel_err(oa[i].cpu().numpy(), oracle.dcn_forward(xa[i], off[i], wa, 1, 1, 1)) <= 1e-4
        assert _rel_err(ob[i].cpu().numpy(), oracle.dcn_forward(xb[i], off[i], wb, 1, 1, 1)) <= 1e-4
    # head shapes (256 -> 256, three levels incl. a ragged last tile), channels-last, fused ReLU
    torch.manual_seed(3)
    sizes = [(40, 40), (20, 20), (7, 9)]
    fa = [torch.randn(1, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last) for h, w in sizes]
    fb = [torch.randn(1, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last) for h, w in sizes]
    of = [torch.randn(1, 18, h, w, device=dev) * 1.5 for h, w in sizes]
    w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
    pa, pb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    sa = deform_conv_forward_multi(fa, of, w1, 1, 1, 1, relu=True)
    sb = deform_conv_forward_multi(fb, of, w2, 1, 1, 1, relu=True)
    for x, y in zip(pa + pb, sa + sb):
        assert x.is_contiguous(memory_format=torch.channels_last) and torch.equal(x, y)
    assert float(pa[0].min()) >= 0.0


def _dcn_torch_reference(x, off, w):
    """Plain PyTorch fp32 DeformConv forward (3x3, stride 1, pad 1, dil 1): explicit bilinear gather + einsum."""
    B, C, H, W = x.shape
    ys, xs = torch.meshgrid(torch.arange(H, device=x.device, dtype=torch.float32),
                            torch.arange(W, device=x.device, dtype=torch.float32), indexing='ij')
    xf = x.reshape(B, C, H * W)
    out = torch.zeros(B, w.size(0), H, W, device=x.device)
    for t in range(9):
        ki, kj = t // 3, t % 3
        h = ys[None] - 1 + ki + off[:, 2 * t]
        ww = xs[None] - 1 + kj + off[:, 2 * t + 1]
        ok = (h > -1) & (ww > -1) & (h < H) & (ww < W)
        h0, w0 = torch.floor(h), torch.floor(ww)
        lh, lw = h - h0, ww - w0
        samp = torch.zeros(B, C, H, W, device=x.device)
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, wx = h0 + dh, w0 + dw
            inb = ok & (hh >= 0) & (hh <= H - 1) & (wx >= 0) & (wx <= W - 1)
            idx = (hh.clamp(0, H - 1) * W + wx.clamp(0, W - 1)).long().reshape(B, 1, H * W).expand(B, C, H * W)
            v = torch.gather(xf, 2, idx).reshape(B, C, H, W)
            samp = samp + v * (wt * inb)[:, None]
        out = out + torch.einsum('oc,bchw->bohw', w[:, :, ki, kj], samp)
    return out


def test_dcn_pair_at_1536_patch_shapes_vs_torch_reference(dev, oracle, dcn_mode):
    """BASELINE configs[4] shapes: a 1536x1536 patch = levels 192^2 .. 12^2 = 49 104 positions, both head DeformConvs in
    one launch (tile table, MT selection and the XCD map at 512 tiles / two rounds).  Checker: a plain PyTorch fp32
    DeformConv, itself pinned to the oracle on a small case first."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
    rng = np.random.RandomState(21)
    xs_, of_, ws_ = rng.normal(size=(1, 32, 9, 11)).astype(np.float32), rng.normal(0, 2.5, size=(1, 18, 9, 11)).astype(np.float32), \
        rng.normal(0, 0.1, size=(16, 32, 3, 3)).astype(np.float32)
    ref_small = _dcn_torch_reference(_t(xs_, dev), _t(of_, dev), _t(ws_, dev)).cpu().numpy()
    assert _rel_err(ref_small, oracle.dcn_forward(xs_, of_, ws_, 1, 1, 1)) <= 1e-5
    torch.manual_seed(5)
    sizes = [1536 // s for s in (8, 16, 32, 64, 128)]
    fa = [torch.randn(1, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
    fb = [torch.randn(1, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
    of = [torch.randn(1, 18, n, n, device=dev) * 2 for n in sizes]
    w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
    pa, pb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
    for lvl in range(5):
        for got, x, w in ((pa[lvl], fa[lvl], w1), (pb[lvl], fb[lvl], w2)):
            want = _dcn_torch_reference(x.contiguous(), of[lvl], w)
            scale = float(want.abs().max())
            assert float((got - want).abs().max()) <= 1e-4 * scale, lvl


@pytest.mark.parametrize("B,channels_last,relu", [(1, False, True), (1, True, False), (2, False, False), (2, True, True),
                                                   (3, False, True)])
def test_dcn_pair_tap_split_launch_at_1024_shapes(dev, B, channels_last, relu, dcn_mode):
    """The configs[1] / configs[2] launches themselves: all five levels of 1024^2 image(s), both head DeformConvs in one
    launch.  B = 1: 228 whole tiles, one round.  B = 2, 3: 456 / 683 tiles do not divide over 256 CUs, so the launch is
    the tap-granular split: XCDs 0-3 / 4-7 take one layer each, their workgroups take the tiles round by round and the
    last, partial round is cut into equal ranges of its (tile, tap) sequence; a cut tile's parts hand their accumulators
    down the chain head -> middle -> tail (3 images: ranges of 2.8 taps would make the chains long: whole tiles).  Checker: the plain PyTorch fp32 DeformConv (pinned to the oracle in the 1536 test); the result must
    also be bitwise reproducible launch to launch (the partial sums are always added in the same order)."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
    torch.manual_seed(6 + B)
    sizes = (128, 64, 32, 16, 8)
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    fa = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=fmt) for n in sizes]
    fb = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=fmt) for n in sizes]
    of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
    w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
    pa, pb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=relu)
    for lvl in range(5):
        for got, x, w in ((pa[lvl], fa[lvl], w1), (pb[lvl], fb[lvl], w2)):
            want = _dcn_torch_reference(x.contiguous(), of[lvl], w)
            if relu:
                want = want.clamp(min=0)
            assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max()), lvl
    from orientedreppoints_amd import _lib
    for it in range(40 if B == 2 else 3):
        # poison the op's scratch (NaN bit patterns) before every launch: a hand-over whose flag could overtake its
        # accumulator stores (the round-3 advisor finding: no s_waitcnt vmcnt(0) before the flag) would add NaNs / stale
        # values from the reused scratch image instead of the predecessor's partial sums
        _lib.workspace(dev, 1).fill_(0xFF)
        qa, qb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=relu)
        for x, y in zip(pa + pb, qa + qb):
            assert torch.equal(x, y), it


def test_postprocess_at_1536_patch_shapes(dev, oracle):
    """configs[4] shapes through decode -> multiclass rotated NMS: 6 720 candidates.  The fused static kernels, the static
    tensor-op path and the reference-shaped dynamic path agree; min-area-rect of every candidate and the rnms keep set
    of the image's class-offset detections agree with the oracle."""
    import sys as _sys
    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import compose_inputs as CI
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.core import rbbox2result_packed
    from orientedreppoints_amd.mmdet_models.registry import build_head
    from orientedreppoints_amd.mmdet_ops import nms_wrapper
    head = build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()
    cls, pts = CI.postprocess_scene(1536, 77)
    cls_t = [torch.from_numpy(c)[None].to(dev) for c in cls]
    pts_t = [torch.from_numpy(p)[None].to(dev) for p in pts]
    assert sum(min(2000, c.shape[1] * c.shape[2]) for c in cls) == 6720
    metas = [CI.img_meta(1536)]
    captured = {}
    orig = nms_wrapper.rnms

    def spy(dets, iou_thr, device_id=None):
        captured['dets'] = dets.detach().cpu().numpy()
        return orig(dets, iou_thr, device_id)
    nms_wrapper.rnms = spy
    try:
        with torch.no_grad():
            dets, labels = head.get_bboxes(cls_t, None, pts_t, None, metas, ConfigDict(test_cfg))[0]
    finally:
        nms_wrapper.rnms = orig
    d = captured['dets']
    assert d.shape[0] > 1000
    want_keep = oracle.rnms(d, 0.4)
    assert dets.shape[0] == min(len(want_keep), 2000)
    # the kept ROWS, not just their number: scores and (class-offset) corners of the oracle's keep set, in the reference's
    # order (ascending index; score-descending top-2000 when more survive)
    got = dets.cpu().numpy()
    lab = labels.cpu().numpy()
    kept = d[want_keep]
    if len(want_keep) > 2000:
        kept = kept[np.argsort(-kept[:, 8], kind="stable")[:2000]]
    assert np.array_equal(got[:, -1], kept[:, 8])
    off = (kept[:, :8] - got[:, 18:26]).astype(np.float64)              # = label * (max coordinate + 1), to fp32 rounding
    nz = lab > 0
    per_class = np.median(off[nz, 0] / lab[nz]) if nz.any() else 0.0
    assert np.max(np.abs(off - (lab * per_class)[:, None]), initial=0.0) <= 1e-3 * max(1.0, float(np.abs(kept[:, :8]).max()))
    for fused in (True, False):
        cfg = ConfigDict(dict(test_cfg)); cfg['fused_postprocess'] = fused
        with torch.no_grad():
            packed = head.get_bboxes(cls_t, None, pts_t, None, metas, cfg, static=True)[0]
        host = packed.cpu().numpy()
        n = int(host[-1, 0])
        assert host[-1, 1] == 0 and n == dets.shape[0]
        assert np.array_equal(host[:n, -1].astype(np.int64), labels.cpu().numpy())
        assert np.array_equal(host[:n, :-1], dets.cpu().numpy())
        assert rbbox2result_packed(packed, 16) is not None


def test_rnms_batched_16_images_as_image_class_segments(dev, oracle):
    """BASELINE.md section 3 / configs[3]: 16 dense images x ~2000 detections as 240 (image x class) segments in ONE
    orp_rnms_batched launch sequence; every segment's keep set equals the oracle's."""
    from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_batched_device
    parts, sizes = [], []
    for im in range(16):
        d, lab = S.gen_dense_scene(2000, 100 + im)
        for c in range(15):
            sel = d[lab == c].astype(np.float32)
            parts.append(sel); sizes.append(len(sel))
    d = np.concatenate(parts, 0)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    keep, num = rnms_batched_device(_t(d, dev), torch.from_numpy(off), max(sizes), 0.4)
    keep, num = keep.cpu().numpy(), num.cpu().numpy()
    assert len(sizes) == 240 and int(num.sum()) > 0
    for s, n in enumerate(sizes):
        want = oracle.rnms(parts[s], 0.4) + off[s]
        assert num[s] == len(want) and np.array_equal(keep[off[s]:off[s] + num[s]], want), s


@pytest.mark.parametrize("dtype,tol", [("float16", 2e-3), ("bfloat16", 1.6e-2)])
def test_dcn_forward_half_precision_vs_fp32_oracle(dev, oracle, dtype, tol):
    """a2 "fp32 / fp16 dispatch" (AT_DISPATCH_FLOATING_TYPES_AND_HALF) / BASELINE configs[4]: the fp16 and bf16 DeformConv
    forward on v_mfma_f32_32x32x16_{f16,bf16} against the fp32 oracle evaluated on the SAME rounded inputs.  Stated
    tolerance: 2e-3 (fp16) / 1.6e-2 (bf16) of the output scale -- the A samples and the outputs are rounded once to the
    storage type (eps 4.9e-4 / 3.9e-3), everything in between is fp32.  DCNv1 multi-level + DCNv2 (mask, bias, ReLU)."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
    dt = getattr(torch, dtype)
    rng = np.random.RandomState(31)
    shapes = [(9, 12), (5, 5), (2, 3)]
    q = lambda a: torch.from_numpy(a).to(dev).to(dt)                     # noqa: E731  (round to the storage type)
    f = lambda t: t.float().cpu().numpy()                                # noqa: E731
    xs = [q(rng.normal(size=(2, 256, h, w)).astype(np.float32)) for h, w in shapes]
    offs = [q(rng.normal(0, 2.0, size=(2, 18, h, w)).astype(np.float32)) for h, w in shapes]
    w = q(rng.normal(0, 0.05, size=(64, 256, 3, 3)).astype(np.float32))
    outs = deform_conv_forward_multi(xs, offs, w, 1, 1, 1)
    for x, o, got in zip(xs, offs, outs):
        assert got.dtype == dt
        want = oracle.dcn_forward(f(x), f(o), f(w), 1, 1, 1)
        assert np.max(np.abs(f(got) - want)) <= tol * np.max(np.abs(want))
    # channels-last in / out, DCNv2 mask + bias + fused ReLU
    xs_cl = [x.contiguous(memory_format=torch.channels_last) for x in xs[:2]]
    masks = [q(rng.uniform(0, 1, size=(2, 9, h, w_)).astype(np.float32)) for h, w_ in shapes[:2]]
    bias = q(rng.normal(0, 0.5, size=(64,)).astype(np.float32))
    outs2 = deform_conv_forward_multi(xs_cl, offs[:2], w, 1, 1, 1, masks=masks, bias=bias, relu=True)
    for x, o, m, got in zip(xs[:2], offs[:2], masks, outs2):
        assert got.is_contiguous(memory_format=torch.channels_last)
        want = np.maximum(oracle.dcn_forward(f(x), f(o), f(w), 1, 1, 1, mask=f(m), bias=f(bias)), 0.0)
        assert np.max(np.abs(f(got) - want)) <= tol * max(1.0, np.max(np.abs(want)))
    # the head's shape: 256 -> 256, against the fp32 MFMA path on the same rounded inputs
    torch.manual_seed(7)
    big = [torch.randn(1, 256, n, n, device=dev).to(dt).contiguous(memory_format=torch.channels_last) for n in (64, 32, 16)]
    boff = [(torch.randn(1, 18, n, n, device=dev) * 2).to(dt) for n in (64, 32, 16)]
    bw = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).to(dt)
    hout = deform_conv_forward_multi(big, boff, bw, 1, 1, 1)
    fout = deform_conv_forward_multi([b.float() for b in big], [o.float() for o in boff], bw.float(), 1, 1, 1)
    for a, b in zip(hout, fout):
        assert float((a.float() - b).abs().max()) <= tol * float(b.abs().max())


def test_dcn_full_size_properties(dev, dcn_mode):
    """BASELINE shapes (all five levels of a 1024^2 image, 256 -> 256, one launch, MT = 3 tiles): with zero offsets the
    DeformConv IS the plain 3x3 convolution (independent implementation: the library's), and with random offsets it is
    linear in its input."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
    torch.manual_seed(4)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
    sizes = (128, 64, 32, 16, 8)
    xs = [torch.randn(1, 256, h, h, device=dev) for h in sizes]
    zero = [torch.zeros(1, 18, h, h, device=dev) for h in sizes]
    with torch.no_grad():
        got = deform_conv_forward_multi(xs, zero, w, 1, 1, 1)
        for g_, x in zip(got, xs):
            want = torch.nn.functional.conv2d(x, w, None, 1, 1)
            assert float((g_ - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))
        offs = [torch.randn(1, 18, h, h, device=dev) * 2.5 for h in sizes]
        x2 = [torch.randn(1, 256, h, h, device=dev) for h in sizes]
        a = deform_conv_forward_multi(xs, offs, w, 1, 1, 1)
        b = deform_conv_forward_multi(x2, offs, w, 1, 1, 1)
        c = deform_conv_forward_multi([p + 2.0 * q for p, q in zip(xs, x2)], offs, w, 1, 1, 1)
        for a_, b_, c_ in zip(a, b, c):
            ref = a_ + 2.0 * b_
            assert float((c_ - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max()))


def test_dcn_v2_and_fused_epilogue_on_mfma_path(dev, oracle, dcn_mode):
    """DCNv2 (mask + bias) and the fused ReLU epilogue on both MFMA kernel generations, against the oracle."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, modulated_deform_conv
    rng = np.random.RandomState(11)
    for B, C, H, W, Cout in ((1, 256, 12, 9, 256), (2, 64, 7, 10, 128)):
        x, off, w = _dcn_case(12, B, C, H, W, Cout)
        m = rng.uniform(0, 1, size=(B, 9, H, W)).astype(np.float32)
        b = rng.normal(size=(Cout,)).astype(np.float32)
        want = oracle.dcn_forward(x, off, w, 1, 1, 1, mask=m, bias=b)
        got = deform_conv_forward_multi([_t(x, dev)], [_t(off, dev)], _t(w, dev), 1, 1, 1, masks=[_t(m, dev)],
                                        bias=_t(b, dev))[0].cpu().numpy()
        assert _rel_err(got, want) <= 1e-4
        got_r = deform_conv_forward_multi([_t(x, dev)], [_t(off, dev)], _t(w, dev), 1, 1, 1, masks=[_t(m, dev)],
                                          bias=_t(b, dev), relu=True)[0].cpu().numpy()
        assert _rel_err(got_r, np.maximum(want, 0)) <= 1e-4
        # the autograd op takes the same path in its forward
        y = modulated_deform_conv(_t(x, dev), _t(off, dev), _t(m, dev), _t(w, dev), _t(b, dev), 1, 1, 1, 1, 1)
        assert _rel_err(y.cpu().numpy(), want) <= 1e-4


def test_dcn_direct_paths(dev, oracle):
    """groups / deformable groups / stride / DCNv2 mask + bias go through the direct kernel."""
    from orientedreppoints_amd.mmdet_ops import deform_conv, modulated_deform_conv, DeformConv
    rng = np.random.RandomState(3)
    x = rng.normal(size=(2, 8, 9, 10)).astype(np.float32)
    w = rng.normal(0, 0.2, size=(6, 4, 3, 3)).astype(np.float32)          # groups = 2
    off = rng.normal(0, 1.5, size=(2, 2 * 18, 5, 5)).astype(np.float32)   # dg = 2, stride 2
    got = deform_conv(_t(x, dev), _t(off, dev), _t(w, dev), 2, 1, 1, 2, 2, 64).cpu().numpy()
    want = oracle.dcn_forward(x, off, w, stride=2, pad=1, dil=1, groups=2, dg=2)
    assert _rel_err(got, want) <= 1e-4
    w2 = rng.normal(0, 0.2, size=(6, 8, 3, 3)).astype(np.float32)
    off2 = rng.normal(0, 1.5, size=(2, 18, 9, 10)).astype(np.float32)
    mask = rng.uniform(0, 1, size=(2, 9, 9, 10)).astype(np.float32)
    bias = rng.normal(size=6).astype(np.float32)
    got = modulated_deform_conv(_t(x, dev), _t(off2, dev), _t(mask, dev), _t(w2, dev), _t(bias, dev), 1, 1, 1, 1, 1)
    want = oracle.dcn_forward(x, off2, w2, mask=mask, bias=bias)
    assert _rel_err(got.cpu().numpy(), want) <= 1e-4
    with pytest.raises(NotImplementedError):
        deform_conv(torch.zeros(1, 8, 4, 4), torch.zeros(1, 18, 4, 4), torch.zeros(8, 8, 3, 3), 1, 1, 1, 1, 1, 64)
    m = DeformConv(64, 64, 3, padding=1).to(dev)
    y = m(torch.randn(1, 64, 2, 2, device=dev), torch.zeros(1, 18, 2, 2, device=dev))   # smaller than the kernel
    assert y.shape == (1, 64, 2, 2)


# ---- convex_giou -----------------------------------------------------------------------------------------------
def test_convex_giou_values_and_grads(dev, oracle, golden_dir):
    """giou: fp64 + - * / only -> 1e-6; gradients: reverse-mode vs the reference's dense Jacobian products -> 1e-4."""
    from orientedreppoints_amd.mmdet_ops import convex_giou
    g = _g(golden_dir, "convex_giou.npz")
    giou, grad = convex_giou(_t(g["pts"], dev), _t(g["gts"], dev))
    assert giou.shape == (480,) and grad.shape == (480, 18)
    assert np.max(np.abs(giou.cpu().numpy() - g["out19"][:, 18])) <= 1e-6
    assert np.max(np.abs(grad.cpu().numpy() - g["out19"][:, :18])) <= 1e-4
    gts = S.gen_gts(3000, 321).astype(np.float32)
    ctr = gts.reshape(-1, 4, 2).mean(1) + np.random.RandomState(5).normal(0, 15, (3000, 2))
    pts = S.gen_pointsets(3000, 322, around=ctr).astype(np.float32)
    want, flags = oracle.convex_giou(pts, gts, return_flags=True)
    giou, grad = convex_giou(_t(pts, dev), _t(gts, dev))
    ok = flags == 0
    assert ok.mean() > 0.95
    assert np.max(np.abs(giou.cpu().numpy()[ok] - want[ok, 18])) <= 1e-6
    assert np.max(np.abs(grad.cpu().numpy()[ok] - want[ok, :18])) <= 1e-4
    assert (np.abs(want[:, :18]).sum(1) > 0).mean() > 0.9        # gradients are not trivially zero


def test_giou_loss_module(dev, oracle):
    from orientedreppoints_amd.mmdet_models.losses import GIoULoss
    gts = S.gen_gts(200, 11).astype(np.float32)
    pts = S.gen_pointsets(200, 12, around=gts.reshape(-1, 4, 2).mean(1)).astype(np.float32)
    pred = _t(pts, dev).requires_grad_(True)
    w = torch.linspace(0.5, 1.0, 200, device=dev)
    loss = GIoULoss(loss_weight=0.375)(pred, _t(gts, dev), w)
    loss.backward()
    want = oracle.convex_giou(pts, gts)
    wn = w.cpu().numpy()
    assert abs(float(loss) - 0.375 * np.mean((1 - want[:, 18]) * wn)) <= 1e-4
    gref = want[:, :18] * wn[:, None]
    bad = (gref > 1).sum(1) > 0
    gref[bad] = 1e-6
    gref = -gref / 200 * 0.375
    assert np.max(np.abs(pred.grad.cpu().numpy() - gref)) <= 1e-5


# ---- assigners / APAA ------------------------------------------------------------------------------------------------
def _apaa(golden_dir):
    return np.load(os.path.join(golden_dir, "apaa_py.npz"))


def test_point_assigner_kernel(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_models.assigners import PointAssigner
    g = _apaa(golden_dir)
    for pn in (1, 3):
        r = PointAssigner(scale=4, pos_num=pn).assign(_t(g["points"], dev), _t(g["gts"], dev), None,
                                                      torch.from_numpy(g["gt_labels"]).to(dev))
        assert np.array_equal(r.gt_inds.cpu().numpy(), g["pa_gt_inds_%d" % pn])          # reference python
        assert np.array_equal(r.labels.cpu().numpy(), g["pa_labels_%d" % pn])
    # one 1024^2 image, 300 gts, against the oracle
    pts = []
    for s in (8, 16, 32, 64, 128):
        f = 1024 // s
        ys, xs = np.meshgrid(np.arange(f) * s, np.arange(f) * s, indexing="ij")
        pts.append(np.stack([xs.ravel(), ys.ravel(), np.full(f * f, s)], 1))
    pts = np.concatenate(pts, 0).astype(np.float32)
    gts = S.gen_gts(300, 4).astype(np.float32)
    r = PointAssigner(scale=4, pos_num=1).assign(_t(pts, dev), _t(gts, dev))
    assert np.array_equal(r.gt_inds.cpu().numpy(), oracle.point_assign(pts, gts, 4, 1))
    r0 = PointAssigner().assign(_t(pts, dev), torch.zeros((0, 8), device=dev))
    assert int(r0.gt_inds.abs().sum()) == 0


def test_max_iou_assigner_kernel(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_models.assigners import MaxIoUAssigner
    g = _apaa(golden_dir)
    a = MaxIoUAssigner(pos_iou_thr=0.1, neg_iou_thr=0.1, min_pos_iou=0, ignore_iof_thr=-1)
    r = a.assign_wrt_overlaps(_t(g["overlaps"], dev), torch.from_numpy(g["gt_labels"]).to(dev))
    assert np.array_equal(r.gt_inds.cpu().numpy(), g["mia_gt_inds"])
    assert np.array_equal(r.max_overlaps.cpu().numpy(), g["mia_max_overlaps"])
    assert np.array_equal(r.labels.cpu().numpy(), g["mia_labels"])
    a2 = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.3, min_pos_iou=0.2, ignore_iof_thr=-1)
    assert np.array_equal(a2.assign_wrt_overlaps(_t(g["overlaps"], dev)).gt_inds.cpu().numpy(), g["mia2_gt_inds"])
    # full assign(): convex_iou + assignment on the device, against oracle(convex_iou) + oracle(assign)
    r = a.assign(_t(g["psets"], dev), _t(g["gts"], dev))
    ov = oracle.convex_iou(g["psets"], g["gts"])
    gi, mo = oracle.max_iou_assign(ov, 0.1, 0.1, 0.0, True)
    assert np.array_equal(r.gt_inds.cpu().numpy(), gi)
    # NaN overlaps behave as torch.max does (NaN wins): row stays -1
    ovn = g["overlaps"].copy(); ovn[3, 7] = np.nan
    rn = a.assign_wrt_overlaps(_t(ovn, dev)).gt_inds.cpu().numpy()
    gin, _ = oracle.max_iou_assign(np.ascontiguousarray(ovn.T), 0.1, 0.1, 0.0, True)
    assert np.array_equal(rn, gin) and rn[7] == -1


def test_apaa_feature_dissimilarity_and_selection(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_ops import apaa
    g = _apaa(golden_dir)
    # sampling + cosine in one kernel vs (oracle sample_points -> oracle dissimilarity) and vs torch grid_sample
    rng = np.random.RandomState(3)
    feats = [rng.normal(size=(2, 64, h, h)).astype(np.float32) for h in (32, 16, 8)]
    strides = [8, 16, 32]
    P = 500
    lvl = rng.randint(0, 3, P).astype(np.int32); img = rng.randint(0, 2, P).astype(np.int32)
    pts = rng.uniform(-20, 276, (P, 18)).astype(np.float32)
    got = apaa.apaa_feature_dissimilarity([_t(f, dev) for f in feats], strides, _t(pts, dev),
                                          torch.from_numpy(img).to(dev), torch.from_numpy(lvl).to(dev)).cpu().numpy()
    want = np.empty(P, np.float32)
    for i in range(P):
        s = oracle.sample_points(feats[lvl[i]][img[i]], strides[lvl[i]], pts[i:i + 1])
        want[i] = oracle.feature_dissimilarity(s)[0]
    assert np.max(np.abs(got - want)) <= 1e-4
    # selection: reference python golden + a larger random case against the oracle
    pos = g["qa_pos_inds"]
    bounds = np.cumsum([0, 1024, 256, 64, 16, 4])
    plvl = (np.searchsorted(bounds, pos, side="right") - 1).astype(np.int32)
    keep = apaa.apaa_select(_t(g["qa_out"], dev), torch.from_numpy(g["sel_pos_gt_inds"]).to(dev),
                            torch.from_numpy(plvl).to(dev), int(g["sel_pos_gt_inds"].max()), 5).cpu().numpy()
    assert np.array_equal(keep, g["sel_label"][pos] > 0) and keep.sum() == int(g["sel_num_pos"])
    q = rng.uniform(0, 5, 6000).astype(np.float32)
    gt = rng.randint(1, 200, 6000).astype(np.int64); lv = rng.randint(0, 5, 6000).astype(np.int32)
    keep = apaa.apaa_select(_t(q, dev), torch.from_numpy(gt).to(dev), torch.from_numpy(lv).to(dev), 199, 5).cpu().numpy()
    assert np.array_equal(keep, oracle.apaa_select(q, gt, lv, 199).astype(bool))


def test_points_quality_assessment_vs_reference_python(dev, golden_dir, oracle):
    """Q of every positive (focal + GIoU x2 + chamfer x2) against the reference's own points_quality_assessment run on
    CPU through the oracle ops.  This golden fed a ready-made [N,9,32] feature tensor, so the feature term is formed
    here with the same formula; the whole Q INCLUDING the sampled-feature kernel is compared with the reference's
    loss() in tests/test_gpu_compose.py::test_head_loss_vs_reference_python."""
    import types
    from orientedreppoints_amd.mmdet_models import orientedreppoints_head_train as T
    from orientedreppoints_amd.mmdet_models.losses import FocalLoss, GIoULoss
    g = _apaa(golden_dir)
    head = types.SimpleNamespace(num_points=9, point_strides=[8, 16, 32, 64, 128],
                                 loss_cls=FocalLoss(), loss_rbox_refine=GIoULoss())
    pos = torch.from_numpy(g["qa_pos_inds"]).to(dev)
    N = g["psets"].shape[0]
    label = torch.from_numpy(g["mia_labels"]).to(dev)
    rbox_w = torch.zeros(N, device=dev); rbox_w[pos] = 1.0
    # the golden used a [N,9,32] feature tensor directly: take that term from the oracle-equivalent torch formula
    f = _t(g["qa_pfeat"], dev)[pos]
    mean = f.mean(1, keepdim=True)
    u = f / f.norm(dim=2, keepdim=True).clamp(min=1e-2); v = mean / mean.norm(dim=2, keepdim=True).clamp(min=1e-2)
    feat_term = (1 - torch.nn.functional.cosine_similarity(u, v, dim=2, eps=1e-6)).max(1)[0]
    import orientedreppoints_amd.mmdet_ops.apaa as apaa_mod
    orig = apaa_mod.apaa_feature_dissimilarity
    apaa_mod.apaa_feature_dissimilarity = lambda *a, **k: feat_term
    try:
        q = T.points_quality_assessment(head, None, 0, torch.zeros(N, dtype=torch.int32, device=dev),
                                        _t(g["qa_cls_score"], dev), _t(g["psets"], dev), _t(g["qa_pts_refine"], dev),
                                        label, _t(g["qa_rbbox_gt"], dev), torch.ones(N, device=dev), rbox_w, pos)
    finally:
        apaa_mod.apaa_feature_dissimilarity = orig
    # bar 1e-4 on EVERY positive (rounds 3-5 exempted "provable min-area-rect ties": the kernel's cosine differed from the host
    # library's in the last ulp and the first-strict-minimum rule then picked another edge direction; round 6: same cosine bits)
    posn = g["qa_pos_inds"]
    tie = np.minimum(oracle.minarearect_margin(g["psets"][posn]), oracle.minarearect_margin(g["qa_pts_refine"][posn])) < 1e-5
    d = np.abs(q.cpu().numpy() - g["qa_out"])
    assert np.max(d, initial=0.0) <= 1e-4 and np.mean(d) <= 2e-6
    import conftest
    conftest.REPORT.append("a15, points_quality_assessment golden: all %d quality values within 1e-4 (largest difference %.2e); %d of "
                           "them sit on min-area-rect ties, largest difference there %.2e"
                           % (tie.size, float(np.max(d, initial=0.0)), int(tie.sum()), float(np.max(d[tie], initial=0.0))))
    sp = T.sampling_points(_t(g["gts"], dev), 10).cpu().numpy()
    assert np.max(np.abs(sp - g["sampling_points"])) <= 1e-5


# ---- deformable convolution backward ------------------------------------------------------------------------------------
def test_dcn_backward(dev, oracle):
    from orientedreppoints_amd.mmdet_ops import deform_conv, modulated_deform_conv
    x, off, w = _dcn_case(5, 2, 64, 10, 12, 64)
    go = np.random.RandomState(6).normal(size=(2, 64, 10, 12)).astype(np.float32)
    tx, toff, tw = (_t(a, dev).requires_grad_(True) for a in (x, off, w))
    out = deform_conv(tx, toff, tw, 1, 1, 1, 1, 1, 64)
    out.backward(_t(go, dev))
    gi, goff, gw = oracle.dcn_backward(x, off, w, go)
    for got, want in ((tx.grad, gi), (toff.grad, goff), (tw.grad, gw)):
        assert _rel_err(got.cpu().numpy(), want) <= 1e-4
    # DCNv2 (8 -> 4 channels: the column-formulation route) with a zero-initialised bias: all five gradients of sum(y)
    # against oracle.dcn_v2_backward (pinned to the reference's modulated kernels in tests/test_oracle_vs_ref.py)
    rng = np.random.RandomState(7)
    x2n = rng.normal(size=(1, 8, 6, 6)).astype(np.float32)
    off2n = rng.normal(0, 1, size=(1, 18, 6, 6)).astype(np.float32)
    m2n = rng.uniform(0.2, 1, size=(1, 9, 6, 6)).astype(np.float32)
    w2n = rng.normal(0, 0.2, size=(4, 8, 3, 3)).astype(np.float32)
    x2, off2, m2, w2 = (_t(a, dev).requires_grad_(True) for a in (x2n, off2n, m2n, w2n))
    b2 = torch.zeros(4, device=dev, requires_grad=True)
    y = modulated_deform_conv(x2, off2, m2, w2, b2, 1, 1, 1, 1, 1)
    y.sum().backward()
    want = oracle.dcn_v2_backward(x2n, off2n, m2n, w2n, np.ones((1, 4, 6, 6), np.float32))
    for name, t, c in zip(("input", "offset", "mask", "weight", "bias"), (x2, off2, m2, w2, b2), want):
        assert _rel_err(t.grad.cpu().numpy(), c) <= 1e-4, name
    assert np.allclose(b2.grad.cpu().numpy(), 36.0)


def test_dcn_backward_mfma_vs_oracle(dev, oracle):
    """orp_dcn_backward_multi (two MFMA implicit GEMMs, no column buffer; deform_conv_cuda.cpp:262-488) against the oracle's
    column formulation (pinned to the reference's col2im / col2im_coord kernels in tests/test_oracle_vs_ref.py): <= 1e-4
    of each gradient's scale.  Multi-level call incl. offsets far outside the map, tiles that straddle images, and the
    autograd route (DeformConvFunction.backward) at the head's 256 -> 256 channels."""
    from orientedreppoints_amd.mmdet_ops import deform_conv, deform_conv_backward as bw
    shapes = [(10, 12), (7, 9), (3, 3), (1, 2)]
    cases = [_dcn_case(20 + i, 2, 256, h, w, 256, std_off=(2.0 if i != 1 else 6.0)) for i, (h, w) in enumerate(shapes)]
    w = cases[0][2]
    gos = [np.random.RandomState(30 + i).normal(size=(2, 256, h, ww)).astype(np.float32) for i, (h, ww) in enumerate(shapes)]
    # a regression branch only has gradient at its positive points: level 0 keeps three non-zero positions (two of them
    # in one 32-position chunk, one in the second image), level 2 is all zero -> chunks skipped through the active list
    keep = np.zeros((2, 1, 10, 12), np.float32)
    keep[0, 0, 2, 3] = keep[0, 0, 2, 9] = keep[1, 0, 7, 11] = 1.0
    gos[0] *= keep
    gos[2] *= 0.0
    assert bw.mfma_ok(_t(w, dev), 1, 1)
    gis, goffs, gw = bw.backward_mfma([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev),
                                      [_t(g, dev) for g in gos], (1, 1), (1, 1), (1, 1))
    want_gw = np.zeros_like(w)
    for c, g, gi, goff in zip(cases, gos, gis, goffs):
        wi, woff, wgw = oracle.dcn_backward(c[0], c[1], w, g)
        want_gw += wgw
        assert _rel_err(gi.cpu().numpy(), wi) <= 1e-4
        assert _rel_err(goff.cpu().numpy(), woff) <= 1e-4
        if not g.any():
            assert not gi.any() and not goff.any()
    assert _rel_err(gw.cpu().numpy(), want_gw) <= 1e-4
    # deterministic: the same call twice gives the same grad_weight bits (fixed-order split reduction)
    gw2 = bw.backward_mfma([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev),
                           [_t(g, dev) for g in gos], (1, 1), (1, 1), (1, 1), need_input=False)[2]
    assert torch.equal(gw, gw2)
    # autograd route, and agreement with the column formulation at a larger map
    x, off, w2 = _dcn_case(41, 2, 256, 32, 32, 256, std_off=3.0)
    go = np.random.RandomState(42).normal(size=(2, 256, 32, 32)).astype(np.float32)
    grads = {}
    for use in (True, False):
        bw.USE_MFMA = use
        try:
            tx, toff, tw = (_t(a, dev).requires_grad_(True) for a in (x, off, w2))
            deform_conv(tx, toff, tw, 1, 1, 1, 1, 1, 64).backward(_t(go, dev))
            grads[use] = [t.grad.cpu().numpy() for t in (tx, toff, tw)]
        finally:
            bw.USE_MFMA = True
    for a, b in zip(grads[True], grads[False]):
        assert _rel_err(a, b) <= 1e-4
    wi, woff, wgw = oracle.dcn_backward(x, off, w2, go)
    for a, b in zip(grads[True], (wi, woff, wgw)):
        assert _rel_err(a, b) <= 1e-4


def test_dcn_backward_input_without_atomics_is_bitwise_reproducible(dev, oracle):
    """grad_input / grad_offset of orp_dcn_backward_multi come from the region-owner formulation (kernel A2 of
    csrc/orp_dcn_bwd_mfma.hip: no atomics, fixed summation order): three calls return the SAME BITS, the values match the
    oracle's column formulation to 1e-4 of scale, on maps whose sizes are not multiples of the 8 x 8 regions, with
    offsets that throw samples across region / image borders and far outside the map, and with B = 3."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw
    shapes = [(19, 21), (8, 8), (5, 13)]
    cases = [_dcn_case(60 + i, 3, 256, h, w, 256, std_off=(1.5, 9.0, 30.0)[i]) for i, (h, w) in enumerate(shapes)]
    w = cases[0][2]
    gos = [np.random.RandomState(70 + i).normal(size=(3, 256, h, ww)).astype(np.float32) for i, (h, ww) in enumerate(shapes)]
    gos[1][1] = 0.0                                                       # one image without gradient
    runs = []
    for _ in range(3):
        gis, goffs, _ = bw.backward_mfma([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev),
                                         [_t(g, dev) for g in gos], (1, 1), (1, 1), (1, 1), need_weight=False)
        runs.append(([g.clone() for g in gis], [g.clone() for g in goffs]))
        junk = torch.full((1 << 24,), float("nan"), device=dev); del junk   # recycle the workspace neighbourhood
    for gis, goffs in runs[1:]:
        for a, b in zip(gis + goffs, runs[0][0] + runs[0][1]):
            assert torch.equal(a, b), "grad_input / grad_offset bits differ between identical calls"
    for c, g, gi, goff in zip(cases, gos, runs[0][0], runs[0][1]):
        wi, woff, _ = oracle.dcn_backward(c[0], c[1], w, g)
        assert _rel_err(gi.cpu().numpy(), wi) <= 1e-4
        assert _rel_err(goff.cpu().numpy(), woff) <= 1e-4
        assert np.isfinite(gi.cpu().numpy()).all()
    # ORP_DCN_BWD_SPARSE (the head's refinement branch): same values from the atomic scatter
    gis, goffs, _ = bw.backward_mfma([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev),
                                     [_t(g, dev) for g in gos], (1, 1), (1, 1), (1, 1), need_weight=False, sparse_grad=True)
    for a, b in zip(gis + goffs, runs[0][0] + runs[0][1]):
        assert _rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 1e-4


def test_dcn_backward_fp16_pieces_range_scaling(dev, oracle):
    """The backward's grad_input / grad_offset contraction carries grad_out and W as two fp16 pieces after a power-of-two range
    scaling taken from max |grad_out| and max |W| of the CALL (csrc/orp_dcn_bwd_mfma.hip, round 5).  The result must not depend on
    the magnitudes: gradients scaled by 2^-60 / 2^40 and weights by 2^20 / 2^-30 give the correspondingly scaled results BIT FOR BIT
    on the fixed-order route (powers of two commute with every rounding of the path) and to 1e-4 on the atomic route, and an
    all-zero gradient gives exact zeros (range word 0: scale 1)."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw
    shapes = [(9, 11), (4, 5)]
    cases = [_dcn_case(90 + i, 2, 256, h, w, 256, std_off=2.5) for i, (h, w) in enumerate(shapes)]
    w = cases[0][2]
    gos = [np.random.RandomState(95 + i).normal(size=(2, 256, h, ww)).astype(np.float32) for i, (h, ww) in enumerate(shapes)]

    def run(gscale, wscale, sparse=False):
        gis, goffs, _ = bw.backward_mfma([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w * np.float32(wscale), dev),
                                         [_t(g * np.float32(gscale), dev) for g in gos], (1, 1), (1, 1), (1, 1), need_weight=False,
                                         sparse_grad=sparse)
        return [g.cpu().numpy() for g in gis], [g.cpu().numpy() for g in goffs]
    base_i, base_o = run(1.0, 1.0)
    for c, g, gi, goff in zip(cases, gos, base_i, base_o):
        wi, woff, _ = oracle.dcn_backward(c[0], c[1], w, g)
        assert _rel_err(gi, wi) <= 1e-4 and _rel_err(goff, woff) <= 1e-4
    for gscale, wscale in ((2.0 ** -60, 2.0 ** 20), (2.0 ** 40, 2.0 ** -30)):
        for sparse in (False, True):
            gi2, go2 = run(gscale, wscale, sparse)
            k = np.float32(gscale) * np.float32(wscale)
            for a, b in zip(gi2 + go2, base_i + base_o):
                if sparse:
                    assert _rel_err(a, b * k) <= 1e-4              # (atomics: another summation order)
                else:
                    assert np.array_equal(a, b * k), "a power-of-two scaling of grad_out / W changed the result's bits"
    # an all-zero gradient: range word 0 -> scale 1, exact zeros, no NaN
    gis, goffs, _ = bw.backward_mfma([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev),
                                     [torch.zeros_like(_t(g, dev)) for g in gos], (1, 1), (1, 1), (1, 1), need_weight=False)
    for t in gis + goffs:
        assert not t.cpu().numpy().any()


def test_dcn_backward_weight_gradient_on_the_16bit_pipe(dev, oracle):
    """grad_weight of orp_dcn_backward_multi runs on the 16-bit matrix pipe since round 6 (dcn_bwd_weight16_kernel: sampled columns and
    grad_out as two fp16 pieces after power-of-two range scalings taken from max |x| and max |grad_out| of the CALL).  Against the
    exact-fp32 MFMA kernel of the same library (ORP_DCN_BWD_W16=0, in a subprocess) the difference stays at the pieces' 2^-22 level;
    magnitudes do not matter (inputs scaled by 2^30 / 2^-40, gradients by 2^-50 / 2^25: the same bits, scaled); an all-zero gradient
    gives exact zeros; the weight-only call returns the bits of the combined call; a level whose last chunk is partial and a
    single-position level are in the launch."""
    import os, subprocess, sys, tempfile
    from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw
    if os.environ.get("ORP_DCN_BWD_W16", "1") == "0" or os.environ.get("ORP_DCN_BWD_SPLIT", "1") == "0":
        pytest.skip("the 16-bit weight-gradient kernel is switched off in this environment")
    shapes = [(9, 11), (4, 5), (1, 1)]
    cases = [_dcn_case(190 + i, 2, 256, h, w, 256, std_off=2.5) for i, (h, w) in enumerate(shapes)]
    w = cases[0][2]
    gos = [np.random.RandomState(195 + i).normal(size=(2, 256, h, ww)).astype(np.float32) for i, (h, ww) in enumerate(shapes)]

    def run(xscale=1.0, gscale=1.0, need_input=True, zero=False):
        out = bw.backward_mfma([_t(c[0] * np.float32(xscale), dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev),
                               [_t(g * np.float32(0.0 if zero else gscale), dev) for g in gos], (1, 1), (1, 1), (1, 1), need_input=need_input)
        return out[2].cpu().numpy()
    base = run()
    want = np.zeros_like(w)
    for c, g in zip(cases, gos):
        want += oracle.dcn_backward(c[0], c[1], w, g)[2]
    assert _rel_err(base, want) <= 1e-4
    assert np.array_equal(run(need_input=False), base), "the weight-only call and the combined call differ"
    for xs, gs in ((2.0 ** 30, 2.0 ** -50), (2.0 ** -40, 2.0 ** 25)):
        got = run(xs, gs)
        assert np.array_equal(got, base * (np.float32(xs) * np.float32(gs))), "a power-of-two scaling of x / grad_out changed grad_weight's bits"
    assert not run(zero=True).any()
    # the exact-fp32 kernel of the same build, same inputs
    with tempfile.TemporaryDirectory() as tmp:
        np.savez(os.path.join(tmp, "in.npz"), w=w, **{"x%d" % i: c[0] for i, c in enumerate(cases)}, **{"o%d" % i: c[1] for i, c in enumerate(cases)},
                 **{"g%d" % i: g for i, g in enumerate(gos)})
        code = ("import numpy as np, torch, sys; sys.path.insert(0, %r)\n"
                "from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw\n"
                "d = np.load(%r); dev = torch.device('cuda:0'); t = lambda a: torch.from_numpy(a).to(dev)\n"
                "n = %d\n"
                "gw = bw.backward_mfma([t(d['x%%d' %% i]) for i in range(n)], [t(d['o%%d' %% i]) for i in range(n)], t(d['w']),\n"
                "                      [t(d['g%%d' %% i]) for i in range(n)], (1, 1), (1, 1), (1, 1))[2]\n"
                "np.save(%r, gw.cpu().numpy())\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(tmp, "in.npz"),
                                                        len(cases), os.path.join(tmp, "gw.npy"))
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORP_DCN_BWD_W16="0"), stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:]
        exact = np.load(os.path.join(tmp, "gw.npy"))
    assert not np.array_equal(exact, base), "ORP_DCN_BWD_W16=0 did not switch kernels"
    assert _rel_err(base, exact) <= 4e-6


def test_dcn_v2_backward_on_mfma_path_vs_oracle(dev, oracle):
    """modulated_deform_conv backward at the head's 256 -> 256 channels through orp_dcn_backward_multi_ex (the modulation
    scalar rides in the sample weights of both MFMA GEMMs; grad_mask = G . sampled value) against oracle.dcn_v2_backward =
    the column formulation of deform_conv_cuda.cpp:569-685 over the oracle's modulated kernels, which
    tests/test_oracle_vs_ref.py pins to the reference's own modulated_deformable_{im2col,col2im,col2im_coord}_gpu_kernel
    (deform_conv_cuda_kernel.cu:570-767).  ALL FIVE gradients within 1e-4 of their scale, the forward too; the repo's own
    column-formulation route (groups / deformable groups fall back to it) is held to the same oracle."""
    from orientedreppoints_amd.mmdet_ops import modulated_deform_conv, deform_conv_backward as bw
    rng = np.random.RandomState(21)
    x, off, w = _dcn_case(81, 2, 256, 11, 13, 256, std_off=2.5)
    m = rng.uniform(0.0, 1.0, size=(2, 9, 11, 13)).astype(np.float32)
    b = rng.normal(size=(256,)).astype(np.float32)
    go = rng.normal(size=(2, 256, 11, 13)).astype(np.float32)
    want_y = oracle.dcn_v2_forward(x, off, m, w, b)
    want = oracle.dcn_v2_backward(x, off, m, w, go)
    names = ("input", "offset", "mask", "weight", "bias")
    for use in (True, False):
        bw.USE_MFMA = use
        try:
            ts = [_t(a, dev).requires_grad_(True) for a in (x, off, m, w, b)]
            y = modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], ts[4], 1, 1, 1, 1, 1)
            y.backward(_t(go, dev))
            assert _rel_err(y.detach().cpu().numpy(), want_y) <= 1e-4
            for name, t, c in zip(names, ts, want):
                assert _rel_err(t.grad.cpu().numpy(), c) <= 1e-4, (name, "mfma" if use else "column")
        finally:
            bw.USE_MFMA = True
    # mask == 1 reduces to DCNv1: the oracle's DCNv1 backward
    ones = np.ones_like(m)
    ts = [_t(a, dev).requires_grad_(True) for a in (x, off, ones, w)]
    modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], None, 1, 1, 1, 1, 1).backward(_t(go, dev))
    wi, woff, wgw = oracle.dcn_backward(x, off, w, go)
    for a, c in zip((ts[0].grad, ts[1].grad, ts[3].grad), (wi, woff, wgw)):
        assert _rel_err(a.cpu().numpy(), c) <= 1e-4


@pytest.mark.parametrize("stride,pad,dil,dg", [(2, 1, 1, 1), (1, 2, 2, 1), (1, 1, 1, 2)])
def test_dcn_v2_general_configs_vs_oracle(dev, oracle, stride, pad, dil, dg):
    """The ModulatedDeformConv configurations a ResNet `dcn=dict(modulated=True, ...)` stage uses
    (mmdet/models/backbones/resnet.py:140-161: stride-2 conv2 of a stage's first block, dilation, deformable_groups > 1)
    forward + all five gradients against oracle.dcn_v2_forward / dcn_v2_backward (pinned to the reference kernels)."""
    from orientedreppoints_amd.mmdet_ops import modulated_deform_conv
    rng = np.random.RandomState(200 + stride + 3 * dil + 7 * dg)
    B, C, H, W, Co = 2, 16, 13, 10, 24
    x = rng.normal(size=(B, C, H, W)).astype(np.float32)
    Ho, Wo = oracle._odim(H, pad, dil, 3, stride), oracle._odim(W, pad, dil, 3, stride)
    off = rng.normal(0, 2.0, size=(B, dg * 18, Ho, Wo)).astype(np.float32)
    m = rng.uniform(0, 1, size=(B, dg * 9, Ho, Wo)).astype(np.float32)
    w = rng.normal(0, 0.2, size=(Co, C, 3, 3)).astype(np.float32)
    b = rng.normal(size=(Co,)).astype(np.float32)
    go = rng.normal(size=(B, Co, Ho, Wo)).astype(np.float32)
    ts = [_t(a, dev).requires_grad_(True) for a in (x, off, m, w, b)]
    y = modulated_deform_conv(ts[0], ts[1], ts[2], ts[3], ts[4], stride, pad, dil, 1, dg)
    y.backward(_t(go, dev))
    assert _rel_err(y.detach().cpu().numpy(), oracle.dcn_v2_forward(x, off, m, w, b, stride, pad, dil, dg)) <= 1e-4
    want = oracle.dcn_v2_backward(x, off, m, w, go, stride, pad, dil, dg)
    for name, t, c in zip(("input", "offset", "mask", "weight", "bias"), ts, want):
        assert _rel_err(t.grad.cpu().numpy(), c) <= 1e-4, name


@pytest.mark.parametrize("dcn_type,stride", [("DCNv2", 1), ("DCNv2", 2), ("DCN", 1)])
def test_resnet_bottleneck_with_dcn_forward_backward_vs_oracle(dev, oracle, dcn_type, stride):
    """The model route to DCN / DCNv2 (mmdet/models/backbones/resnet.py:118-235 with `dcn=dict(type=...)`): one Bottleneck
    whose conv2 is the offset-predicting (Modulated)DeformConvPack, forward + backward on the GPU (HIP DeformConv kernels)
    against the SAME module on the CPU with the deformable convolution replaced by oracle.dcn_v2_forward / dcn_v2_backward
    (tests/cpu_standins.py; pinned to the reference's kernels in tests/test_oracle_vs_ref.py): output, the input gradient
    and every parameter gradient (incl. conv_offset, which receives the offset AND mask gradients) within 1e-4 of scale."""
    import copy, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_standins as CS
    from orientedreppoints_amd.mmdet_models.resnet import Bottleneck
    import importlib
    dc_mod = importlib.import_module('orientedreppoints_amd.mmdet_ops.deform_conv')
    torch.manual_seed(17)
    down = None
    if stride != 1:
        down = torch.nn.Sequential(torch.nn.Conv2d(64, 64, 1, stride=stride, bias=False), torch.nn.BatchNorm2d(64))
    blk = Bottleneck(64, 16, stride=stride, downsample=down, dcn=dict(type=dcn_type, deformable_groups=1))
    assert type(blk.conv2).__name__ == ("ModulatedDeformConvPack" if dcn_type == "DCNv2" else "DeformConvPack")
    with torch.no_grad():                                             # non-trivial offsets / masks
        blk.conv2.conv_offset.weight.normal_(0, 0.15)
        blk.conv2.conv_offset.bias.normal_(0, 0.5)
        for bn in (blk.bn1, blk.bn2, blk.bn3):
            bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
            bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    blk.eval()                                                        # norm_eval=True, as the DOTA configs train
    cpu = copy.deepcopy(blk)
    gpu = blk.to(dev)
    x = torch.randn(2, 64, 14, 11)
    go = torch.randn(2, 64, (14 - 1) // stride + 1, (11 - 1) // stride + 1)
    xg = x.to(dev).requires_grad_(True)
    yg = gpu(xg)
    yg.backward(go.to(dev))
    xc = x.clone().requires_grad_(True)
    saved = (dc_mod.deform_conv, dc_mod.modulated_deform_conv)
    dc_mod.deform_conv, dc_mod.modulated_deform_conv = CS._CpuDeformConv.apply, CS._CpuModulatedDeformConv.apply
    try:
        yc = cpu(xc)
        yc.backward(go)
    finally:
        dc_mod.deform_conv, dc_mod.modulated_deform_conv = saved
    assert _rel_err(yg.detach().cpu().numpy(), yc.detach().numpy()) <= 1e-4
    assert _rel_err(xg.grad.cpu().numpy(), xc.grad.numpy()) <= 1e-4
    for (n, pg), (_, pc) in zip(gpu.named_parameters(), cpu.named_parameters()):
        assert pg.grad is not None and pc.grad is not None, n
        assert _rel_err(pg.grad.cpu().numpy(), pc.grad.numpy()) <= 1e-4, n
    assert float(gpu.conv2.conv_offset.weight.grad.abs().max()) > 0


@pytest.mark.parametrize("dtype,tol", [("float16", 4e-3), ("bfloat16", 3e-2)])
def test_dcn_backward_half_precision_vs_fp32_oracle(dev, oracle, dtype, tol):
    """fp16 / bf16 DeformConv backward (the reference's AT_DISPATCH_FLOATING_TYPES_AND_HALF branch of deformable_col2im /
    col2im_coord / im2col, deform_conv_cuda_kernel.cu:353,451): tensors in half, arithmetic fp32 MFMA, gradients returned in
    the tensors' type -- against the fp32 oracle evaluated on the SAME rounded inputs: |delta| <= tol of each gradient's scale
    (tol = the type's rounding of the returned gradient plus of the forward's own output tolerance), DCNv1 and DCNv2."""
    from orientedreppoints_amd.mmdet_ops import deform_conv, modulated_deform_conv
    dt = getattr(torch, dtype)
    rng = np.random.RandomState(31)
    x, off, w = _dcn_case(91, 2, 256, 9, 10, 256, std_off=2.0)
    go = rng.normal(size=(2, 256, 9, 10)).astype(np.float32)
    rnd = lambda a: _t(a, dev).to(dt)                                   # noqa: E731
    tx, toff, tw = (rnd(a).requires_grad_(True) for a in (x, off, w))
    tgo = rnd(go)
    deform_conv(tx, toff, tw, 1, 1, 1, 1, 1, 64).backward(tgo)
    assert tx.grad.dtype == dt and toff.grad.dtype == dt and tw.grad.dtype == dt
    f = lambda t: t.detach().float().cpu().numpy()                      # noqa: E731
    wi, woff, wgw = oracle.dcn_backward(f(tx), f(toff), f(tw), f(tgo))
    for name, a, c in (("input", tx.grad, wi), ("offset", toff.grad, woff), ("weight", tw.grad, wgw)):
        assert _rel_err(f(a), c) <= tol, name
    # DCNv2 in half: against the fp32 oracle (pinned to the reference's modulated kernels) on the rounded tensors
    m = rng.uniform(0.1, 1.0, size=(2, 9, 9, 10)).astype(np.float32)
    hs = [rnd(a).requires_grad_(True) for a in (x, off, m, w)]
    modulated_deform_conv(hs[0], hs[1], hs[2], hs[3], None, 1, 1, 1, 1, 1).backward(tgo)
    want = oracle.dcn_v2_backward(f(hs[0]), f(hs[1]), f(hs[2]), f(hs[3]), f(tgo))
    for name, a, c in zip(("input", "offset", "mask", "weight"), hs, want):
        assert a.grad.dtype == dt
        assert _rel_err(f(a.grad), c) <= tol, name


def test_head_training_forward_all_levels_as_one_dcn_node(dev):
    """Training forward of the head with both DeformConvs of all levels as ONE autograd node (pair launch forward, MFMA
    backward over all levels) == the per-level forward_single route on the column-formulation backward: outputs equal,
    every parameter gradient and the input gradients within 1e-4 of their scale."""
    from orientedreppoints_amd.dota_configs import r50_model
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_head
    from orientedreppoints_amd.mmdet_models.orientedreppoints_head import multi_apply
    from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw
    torch.manual_seed(3)
    cfg = dict(r50_model['bbox_head'])
    head = build_head(ConfigDict(cfg)).to(dev)
    head.init_weights()
    head.train()
    with torch.no_grad():
        head.reppoints_pts_init_out.weight.normal_(0, 0.05)          # offsets of a few pixels, some outside the small maps
    feats0 = [torch.randn(2, 256, s, s, device=dev) for s in (24, 12, 6, 3)]
    seeds = None
    results = {}
    for route in ("multi", "multi_fused_gn", "single"):
        head.zero_grad()
        feats = [f.clone().requires_grad_(True) for f in feats0]
        if route == "multi":
            # the towers' GroupNorm on the framework modules, as in the per-level route: this comparison isolates the
            # DeformConv node (the fused GroupNorm is compared below and, op by op, in tests/test_gpu_train_ops.py)
            head._fused_towers_ok = lambda f: False
            try:
                outs = head.forward(feats)
            finally:
                del head._fused_towers_ok
        elif route == "multi_fused_gn":
            outs = head.forward(feats)
        else:
            bw.USE_MFMA = False
            outs = multi_apply(head.forward_single, feats)
        if seeds is None:
            seeds = [[torch.randn_like(o) for o in outs[k]] for k in range(3)]
        try:
            loss = sum((o * s_).sum() for k in range(3) for o, s_ in zip(outs[k], seeds[k]))
            loss.backward()
        finally:
            bw.USE_MFMA = True
        results[route] = ([o.detach().clone() for k in range(3) for o in outs[k]],
                          {n: p.grad.detach().clone() for n, p in head.named_parameters() if p.grad is not None},
                          [f.grad.detach().clone() for f in feats])
    om, gm, fm = results["multi"]
    os_, gs, fs = results["single"]
    for a, b in zip(om, os_):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    assert set(gm) == set(gs) and 'reppoints_cls_conv.weight' in gm and 'reppoints_pts_refine_conv.weight' in gm
    for n in gm:
        assert float((gm[n] - gs[n]).abs().max()) <= 1e-4 * max(1e-6, float(gs[n].abs().max())), n
    for a, b in zip(fm, fs):
        assert float((a - b).abs().max()) <= 1e-4 * max(1e-6, float(b.abs().max()))
    # The same forward with the towers' GroupNorm + ReLU as one autograd node per layer (group_norm_act_train): outputs
    # equal.  Its GRADIENTS are compared op by op in tests/test_gpu_train_ops.py (1e-4 against torch.nn.GroupNorm on the
    # same inputs) and not here: the two forwards agree to ~5e-7, so among the 2.3 M tower activations of this case one
    # or two sit within that distance of zero and take the other side of the ReLU (measured: ONE element of the 24 x 24
    # level) -- a gradient discontinuity that moves single rows of the tower weights' gradients by up to 1e-2 of scale
    # without either route being wrong.
    of, gf, ff = results["multi_fused_gn"]
    for a, b in zip(of, os_):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    assert set(gf) == set(gs)
    for n in gf:
        assert float((gf[n] - gs[n]).norm()) <= 2e-2 * max(1e-6, float(gs[n].norm())), n


def test_dcn_pair_with_fused_1x1_heads_vs_separate_launches(dev):
    """orp_dcn_forward_pair_heads: both DeformConvs + ReLU + the 1x1 output convolution behind each (+ `pts_out_init`
    residual) in ONE launch, against the separate route (pair launch with ReLU, then F.conv2d): <= 1e-5 of the output
    scale -- tiles that straddle images, levels smaller than a tile, offsets outside the map, MT = 2 and 3 tilings."""
    import torch.nn as nn
    import torch.nn.functional as F
    from orientedreppoints_amd.mmdet_ops import DeformConv, deform_conv_forward_pair
    from orientedreppoints_amd.mmdet_ops.deform_conv import deform_conv_forward_pair_heads, pair_heads_ok
    torch.manual_seed(21)
    ca, cb = DeformConv(256, 256, 3, 1, 1).to(dev), DeformConv(256, 256, 3, 1, 1).to(dev)
    ha, hb = nn.Conv2d(256, 15, 1).to(dev), nn.Conv2d(256, 18, 1).to(dev)
    assert pair_heads_ok(ca, cb, ha, hb, torch.zeros(1, 256, 4, 4, device=dev))
    assert not pair_heads_ok(ca, cb, nn.Conv2d(256, 21, 1).to(dev), hb, torch.zeros(1, 256, 4, 4, device=dev))
    for B, sizes in ((2, [(21, 19), (8, 8), (3, 5), (1, 1)]), (1, [(64, 64), (32, 32), (16, 16)])):
        xa = [torch.randn(B, 256, h, w, device=dev) for h, w in sizes]
        xb = [torch.randn(B, 256, h, w, device=dev) for h, w in sizes]
        offs = [torch.randn(B, 18, h, w, device=dev) * (4.0 if i == 1 else 1.5) for i, (h, w) in enumerate(sizes)]
        res = [torch.randn(B, 18, h, w, device=dev) for h, w in sizes]
        with torch.no_grad():
            oa, ob = deform_conv_forward_pair_heads(xa, xb, offs, ca, cb, ha, hb, residuals_b=res)
            da, db = deform_conv_forward_pair(xa, xb, offs, ca.weight, cb.weight, 1, 1, 1, relu=True)
            for i in range(len(sizes)):
                wa = F.conv2d(da[i], ha.weight, ha.bias)
                wb = F.conv2d(db[i], hb.weight, hb.bias) + res[i]
                assert float((oa[i] - wa).abs().max()) <= 1e-5 * max(1.0, float(wa.abs().max()))
                assert float((ob[i] - wb).abs().max()) <= 1e-5 * max(1.0, float(wb.abs().max()))
            o2a, o2b = deform_conv_forward_pair_heads(xa, xb, offs, ca, cb, ha, hb)          # no residual
            assert float((o2b[0] + res[0] - ob[0]).abs().max()) <= 1e-5 * max(1.0, float(ob[0].abs().max()))


def test_conv1x1_multi_vs_library_convolution(dev):
    """orp_conv1x1_multi (the head's 1x1 output convolutions, all levels in one launch, bias / residual / ReLU / `- sub`
    fused in the head's order) against F.conv2d + the separate passes: <= 1e-5 of the output scale (fp32 FMA chain vs the
    library GEMM's accumulation order)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv1x1_multi, conv1x1_ok
    torch.manual_seed(11)
    sizes = [(37, 41), (16, 16), (8, 8), (3, 5), (1, 1)]
    for cout, relu, use_res, use_sub in ((18, False, False, True), (15, False, False, False), (18, False, True, False),
                                         (32, True, True, True), (7, True, False, False)):
        conv = nn.Conv2d(256, cout, 1).to(dev)
        xs = [torch.randn(2, 256, h, w, device=dev) for h, w in sizes]
        res = [torch.randn(2, cout, h, w, device=dev) for h, w in sizes] if use_res else None
        sub = torch.randn(1, cout, 1, 1, device=dev) if use_sub else None
        assert conv1x1_ok(conv, xs[0])
        out = conv1x1_multi(xs, conv, relu=relu, residuals=res, sub=sub)
        ys, zs = out if use_sub else (out, None)
        for i, x in enumerate(xs):
            want = F.conv2d(x, conv.weight, conv.bias)
            if use_res:
                want = want + res[i]
            if relu:
                want = want.relu()
            scale = max(1.0, float(want.abs().max()))
            assert float((ys[i] - want).abs().max()) <= 1e-5 * scale
            if use_sub:
                assert float((zs[i] - (want - sub)).abs().max()) <= 1e-5 * scale
    assert not conv1x1_ok(nn.Conv2d(256, 33, 1).to(dev), xs[0]) and not conv1x1_ok(nn.Conv2d(256, 8, 3).to(dev), xs[0])


# ---- end to end: one training step of the detector ---------------------------------------------------------------------
def test_detector_train_step_and_inference(dev):
    from orientedreppoints_amd.dota_configs import r50_model, train_cfg, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).to(dev)
    model.train()
    B, size = 2, 256
    img = torch.randn(B, 3, size, size, device=dev)
    metas = [dict(img_shape=(size, size, 3), pad_shape=(size, size, 3), scale_factor=1.0, flip=False)] * B
    gts = [_t(S.gen_polys(6, 40 + i, wh=(16, 120))[:, :8] / 4.0, dev) for i in range(B)]
    labels = [torch.randint(1, 16, (6,), device=dev) for _ in range(B)]
    losses = model(img, metas, return_loss=True, gt_bboxes=gts, gt_labels=labels)
    assert set(losses) == {'loss_cls', 'loss_rbox_init', 'loss_rbox_refine', 'loss_spatial_init', 'loss_spatial_refine'}
    total = 0
    for k, v in losses.items():
        vs = v if isinstance(v, (list, tuple)) else [v]
        for t in vs:
            assert torch.isfinite(t).all(), k
            total = total + t.sum()
    assert float(total) > 0
    total.backward()
    g = model.bbox_head.reppoints_cls_conv.weight.grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0
    assert model.bbox_head.reppoints_pts_init_out.weight.grad is not None
    model.eval()
    with torch.no_grad():
        res = model(img[:1], metas[:1], return_loss=False)
    assert len(res) == 15 and all(r.shape[1] in (9, 27) for r in res)   # empty results are [0,9] as in rbbox2result


def test_box_iou_rotated(dev, oracle, golden_dir):
    from orientedreppoints_amd.mmdet_ops import box_iou_rotated
    g = _g(golden_dir, "box_iou_rotated.npz")
    got = box_iou_rotated(_t(g["a"], dev), _t(g["b"], dev)).cpu().numpy()
    assert got.shape == g["iou"].shape and np.array_equal(got, g["iou"], equal_nan=True)
    # BIT-EXACT (round 6: measured 0 of 2.16 M pairs differing from the reference compiled for the host, tests/checks/
    # box_iou_rotated_bits.py; the test held 1e-4 before): the only library calls are double cos / sin rounded to float
    a = S.gen_rboxes(900, 31).astype(np.float32); b = S.gen_rboxes(400, 32).astype(np.float32)
    b[:300, :2] = a[:300, :2] + np.random.RandomState(1).uniform(-8, 8, (300, 2)).astype(np.float32)
    got = box_iou_rotated(_t(a, dev), _t(b, dev)).cpu().numpy()
    want = oracle.box_iou_rotated(a, b)
    assert np.array_equal(got, want, equal_nan=True) and (want > 0).sum() > 3000
    assert box_iou_rotated(torch.zeros((0, 5), device=dev), _t(b, dev)).shape == (0, 400)


# ---- fused normalisation passes (inference): against the stock PyTorch fp32 modules they replace ---------------------
@pytest.mark.parametrize("B,C,G,sizes", [(1, 256, 32, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]),
                                         (2, 64, 32, [(9, 7), (3, 5)]), (1, 32, 4, [(1, 1), (130, 17)])])
def test_groupnorm_relu_multi_vs_torch(dev, B, C, G, sizes):
    from orientedreppoints_amd.mmdet_ops.fused_norm import group_norm_act_multi
    torch.manual_seed(0)
    gn = torch.nn.GroupNorm(G, C).to(dev)
    with torch.no_grad():
        gn.weight.normal_(1.0, 0.3); gn.bias.normal_(0.0, 0.3)
    xs = [torch.randn(B, C, h, w, device=dev) * 3.0 + 1.5 for h, w in sizes]
    for relu in (True, False):
        with torch.no_grad():
            want = [torch.relu(gn(x)) if relu else gn(x) for x in xs]
            got = group_norm_act_multi([x.clone() for x in xs], gn, relu=relu, inplace=True)
        for g_, w_ in zip(got, want):
            assert float((g_ - w_).abs().max()) <= 1e-4


def test_bn_act_vs_torch(dev):
    from orientedreppoints_amd.mmdet_ops.fused_norm import bn_act
    torch.manual_seed(1)
    for B, C, H, W in [(1, 64, 32, 32), (2, 24, 7, 9)]:
        bn = torch.nn.BatchNorm2d(C).to(dev).eval()
        with torch.no_grad():
            bn.weight.normal_(1.0, 0.3); bn.bias.normal_(0, 0.3)
            bn.running_mean.normal_(0, 1.0); bn.running_var.uniform_(0.3, 2.0)
        x = torch.randn(B, C, H, W, device=dev) * 2
        r = torch.randn(B, C, H, W, device=dev)
        with torch.no_grad():
            assert float((bn_act(x.clone(), bn, relu=True) - torch.relu(bn(x))).abs().max()) <= 1e-5
            assert float((bn_act(x.clone(), bn, residual=r, relu=True) - torch.relu(bn(x) + r)).abs().max()) <= 1e-5
            assert float((bn_act(x.clone(), bn, relu=False) - bn(x)).abs().max()) <= 1e-5


def test_bias_act_multi_bit_exact_vs_torch(dev):
    """One launch for all levels == the separate framework passes (bias add, residual add, ReLU, channel subtraction),
    bit for bit: the operation order per element is the same."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import bias_act_multi
    torch.manual_seed(2)
    for B, C, sizes in [(1, 18, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]), (2, 15, [(9, 7), (3, 5), (1, 1)]),
                        (1, 256, [(32, 32), (5, 3)])]:
        xs = [torch.randn(B, C, h, w, device=dev) * 3 for h, w in sizes]
        rs = [torch.randn(B, C, h, w, device=dev) for h, w in sizes]
        bias = torch.randn(C, device=dev)
        sub = torch.randn(1, C, 1, 1, device=dev)
        bb = bias.view(1, -1, 1, 1)
        got = bias_act_multi([x.clone() for x in xs], bias, relu=True)
        for g_, x in zip(got, xs):
            assert torch.equal(g_, torch.relu(x + bb))
        got = bias_act_multi([x.clone() for x in xs], bias, residuals=rs)
        for g_, x, r in zip(got, xs, rs):
            assert torch.equal(g_, (x + bb) + r)
        ys, zs = bias_act_multi([x.clone() for x in xs], bias, sub=sub)
        for y, z, x in zip(ys, zs, xs):
            assert torch.equal(y, x + bb) and torch.equal(z, (x + bb) - sub)
        got = bias_act_multi([x.clone() for x in xs], None, relu=True, residuals=rs)
        for g_, x, r in zip(got, xs, rs):
            assert torch.equal(g_, torch.relu(x + r))


@pytest.mark.parametrize("B,cin,cout,sizes", [(1, 256, 256, [(128, 128), (32, 32), (16, 16), (8, 8)]),
                                              (2, 128, 128, [(7, 5), (13, 9), (1, 1), (31, 33)]),
                                              (3, 384, 64, [(4, 4), (2, 9)]), (1, 64, 64, [(8, 8)])])
def test_small_level_conv3x3_vs_torch(dev, B, cin, cout, sizes):
    """The small-level 3x3 convolution (one exact-fp32 MFMA launch for all small levels) against F.conv2d: same zero
    padding, every level / image / border position; big levels take the library path unchanged."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv3x3_multi
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=True).to(dev)
    with torch.no_grad():
        conv.weight.normal_(0, 0.05)
    xs = [torch.randn(B, cin, h, w, device=dev) for h, w in sizes]
    with torch.no_grad():
        got = conv3x3_multi(xs, conv)
        want = [torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, padding=1) for x in xs]
    for g_, w_ in zip(got, want):
        assert g_.shape == w_.shape
        assert float((g_.double() - w_).abs().max()) <= 1e-4 * max(1.0, float(w_.abs().max()))


@pytest.mark.parametrize("B,cin,cout,sizes,split_k", [(1, 2048, 256, [(32, 32)], True), (1, 256, 256, [(16, 16)], True),
                                                      (2, 256, 256, [(31, 17), (8, 8), (1, 1), (2, 5)], True),
                                                      (1, 256, 128, [(16, 16), (9, 9)], False)])
def test_small_level_conv3x3_stride2_vs_torch(dev, B, cin, cout, sizes, split_k):
    """The FPN's extra levels (3x3, stride 2, pad 1: P6 from the 2048-channel C5, P7 from P6) on the fixed-order HIP
    convolution, with the grid-level K split and its fixed-order sum of partial images: against F.conv2d in fp64, every
    border position, odd sizes; and bitwise reproducible launch to launch (what the library's kernel for these shapes
    is not)."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv3x3_multi
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(cin, cout, 3, stride=2, padding=1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.normal_(0, 0.02)
    xs = [torch.randn(B, cin, h, w, device=dev) for h, w in sizes]
    with torch.no_grad():
        got = conv3x3_multi(xs, conv, split_k=split_k)
        want = [torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, stride=2, padding=1) for x in xs]
        for g_, w_ in zip(got, want):
            assert g_.shape == w_.shape
            assert float((g_.double() - w_).abs().max()) <= 1e-4 * max(1.0, float(w_.abs().max()))
        for _ in range(3):
            for a, b in zip(conv3x3_multi(xs, conv, split_k=split_k), got):
                assert torch.equal(a, b)


def test_inference_step_is_bitwise_reproducible(dev):
    """Backbone -> neck -> head -> decode / NMS of the same image twice: identical bits.  Every HIP kernel of the path sums
    in a fixed order; the library kernels the detector keeps (Winograd / implicit-GEMM convolutions) are deterministic
    for its shapes, except the split-K pick for the FPN's two extra levels, which therefore run on the HIP convolution."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.0)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    img = torch.randn(1, 3, 1024, 1024, device=dev)
    metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)]
    import conftest
    with torch.no_grad():
        # the stages that are this package's own on FIXED inputs: always reproducible, whatever the library does
        c_fix = [t.clone() for t in model.backbone(img)]
        p_ref = [t.clone() for t in model.neck(c_fix)]
        outs_ref = [[t.clone() for t in lv] for lv in head(p_ref)[:3]]
        post_ref = head.get_bboxes(*(tuple(head(p_ref)) + (metas, model.test_cfg, False)), static=True)[0].clone()
        lib_ok = True
        for _ in range(3):
            c_again = model.backbone(img)
            lib_ok = lib_ok and all(torch.equal(a, b) for a, b in zip(c_again, c_fix))
            p_again = model.neck(c_fix)
            for a, b in zip(p_again[3:], p_ref[3:]):
                assert torch.equal(a, b)                   # P6 / P7: the fixed-order HIP convolution + fused GroupNorm
            lib_ok = lib_ok and all(torch.equal(a, b) for a, b in zip(p_again[:3], p_ref[:3]))
            o_again = head(p_ref)
            lib_ok = lib_ok and all(torch.equal(a, b) for la, lb in zip(o_again[:3], outs_ref) for a, b in zip(la, lb))
            packed = head.get_bboxes(*(tuple(o_again) + (metas, model.test_cfg, False)), static=True)[0]
            if all(torch.equal(a, b) for la, lb in zip(o_again[:3], outs_ref) for a, b in zip(la, lb)):
                assert torch.equal(packed, post_ref)       # decode -> selection -> NMS -> packing: HIP kernels only
        if not lib_ok:
            conftest.REPORT.append("inference step: a LIBRARY convolution of backbone / FPN / towers is not bitwise reproducible on this "
                                   "box; the whole-step check was skipped (the HIP stages on fixed inputs were checked)")
            pytest.skip("library convolutions not bitwise reproducible on this box")
        ref = model.simple_test(img, metas)
        assert sum(len(c) for c in ref) > 100
        ref_feats = [f.clone() for f in model.extract_feat(img)]
        for _ in range(3):
            for a, b in zip(model.extract_feat(img), ref_feats):
                assert torch.equal(a, b)
            for a, b in zip(model.simple_test(img, metas), ref):
                assert a.shape == b.shape and np.array_equal(a, b)
    # the extra levels equal the stock modules' (library convolution + framework GroupNorm) to rounding
    with torch.no_grad():
        c = model.backbone(img)
        fused = model.neck(c)
        with torch.enable_grad():
            stock = model.neck([t.detach() for t in c])          # autograd on: the unfused module path
        for a, b in zip(fused, stock):
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))


def test_inference_step_at_1536_is_bitwise_reproducible_in_library_deterministic_mode(dev):
    """configs[4] patch shapes (1536^2): MIOpen's default pick for four backbone convolutions (layer2.0.conv2 stride 2 at 384^2,
    layer4.{0,1,2}.conv2 at 96^2 / 48^2; tests/checks/determinism_modules.py) is a split-K solver that accumulates with
    atomics.  With torch.backends.cudnn.deterministic (MIOpen's deterministic attribute) the whole step -- features and
    detections -- is identical bits run to run; every HIP kernel of the path sums in a fixed order anyway.  bench.py
    switches that mode on by itself when its reproducibility probe fails."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.0)
    img = torch.randn(1, 3, 1536, 1536, device=dev)
    metas = [dict(img_shape=(1536, 1536, 3), pad_shape=(1536, 1536, 3), scale_factor=1.0, flip=False)]
    saved = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        with torch.no_grad():
            ref_feats = [f.clone() for f in model.extract_feat(img)]
            ref = model.simple_test(img, metas)
            assert sum(len(c) for c in ref) > 100
            for _ in range(4):
                for a, b in zip(model.extract_feat(img), ref_feats):
                    assert torch.equal(a, b)
                for a, b in zip(model.simple_test(img, metas), ref):
                    assert a.shape == b.shape and np.array_equal(a, b)
    finally:
        torch.backends.cudnn.deterministic = saved


def test_groupnorm_channels_last_output_and_head_handover(dev):
    """orp_groupnorm_act_multi_nhwc: the normalisation's second pass written transposed (channels-last), alone or next to the
    NCHW result -- the same bits as the plain launch pair, levels whose sizes are not multiples of the 32 x 32 tile, B = 2;
    and the head's inference forward with the towers' last layer handed to the DeformConv pair launch channels-last (no
    transposition kernel) == the NCHW hand-over, bit for bit."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import group_norm_act_multi
    torch.manual_seed(4)
    gn = torch.nn.GroupNorm(32, 256).to(dev)
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5); gn.bias.normal_(0, 0.2)
    xs = [torch.randn(2, 256, h, w, device=dev) * 3 + 1 for h, w in ((37, 41), (16, 16), (5, 3), (2, 2))]
    want = group_norm_act_multi([x.clone() for x in xs], gn, relu=True, inplace=True)
    only = group_norm_act_multi([x.clone() for x in xs], gn, relu=True, inplace=True, nhwc='only')
    both_nchw, both_cl = group_norm_act_multi([x.clone() for x in xs], gn, relu=True, inplace=True, nhwc='both')
    for w_, o, bn, bc in zip(want, only, both_nchw, both_cl):
        assert o.is_contiguous(memory_format=torch.channels_last) and bc.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(o, w_) and torch.equal(bc, w_) and torch.equal(bn, w_) and bn.is_contiguous()
    ref = torch.nn.functional.relu(gn(xs[0]))
    assert float((want[0] - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # the head
    from orientedreppoints_amd.dota_configs import r50_model
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.registry import build_head
    head = build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()
    with torch.no_grad():
        head.reppoints_pts_init_out.weight.normal_(0, 0.05)
        head.split_towers = False           # the library-convolution towers (the channels-last towers: test_gpu_conv_split.py)
        feats = [torch.randn(1, 256, n, n, device=dev) for n in (40, 20, 10, 5, 3)]
        outs = {}
        for flag in (True, False):
            head.nhwc_handover = flag
            o = head(feats)
            outs[flag] = [[t.clone() for t in lv] for lv in o[:3]]
    for la, lb in zip(outs[True], outs[False]):
        for a, b in zip(la, lb):
            assert a.is_contiguous() and torch.equal(a, b)


def test_multi_launch_ops_with_per_tensor_parameters(dev):
    """group_norm_act_multi / conv3x3_multi with one module PER TENSOR (the *_ex entry points): each tensor must get
    its own affine parameters / weights."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv3x3_multi, group_norm_act_multi
    torch.manual_seed(5)
    gns = [torch.nn.GroupNorm(32, 128).to(dev) for _ in range(3)]
    convs = [torch.nn.Conv2d(128, 64, 3, padding=1, bias=False).to(dev) for _ in range(3)]
    with torch.no_grad():
        for g in gns:
            g.weight.normal_(1.0, 0.3); g.bias.normal_(0.0, 0.3)
    xs = [torch.randn(2, 128, h, w, device=dev) for h, w in ((16, 16), (8, 8), (5, 7))]
    with torch.no_grad():
        got = group_norm_act_multi([x.clone() for x in xs], gns, relu=True)
        for g_, x, m in zip(got, xs, gns):
            assert float((g_ - torch.relu(m(x))).abs().max()) <= 1e-4
        got = conv3x3_multi(xs, convs)
        for g_, x, c in zip(got, xs, convs):
            want = torch.nn.functional.conv2d(x.double(), c.weight.double(), None, padding=1)
            assert float((g_.double() - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max()))


def test_detector_fused_inference_matches_stock_modules(dev):
    """The fused inference forward (GroupNorm+ReLU launch pairs, folded BatchNorm) against the same model run through
    the stock PyTorch modules (the autograd-capable per-level forward)."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.uniform_(0.5, 1.5); m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
    img = torch.randn(1, 3, 256, 256, device=dev)
    with torch.no_grad():
        fused = model.bbox_head(model.extract_feat(img))
    with torch.enable_grad():
        stock = model.bbox_head(model.extract_feat(img))
    for a_list, b_list in zip(fused[:3], stock[:3]):
        for a, b in zip(a_list, b_list):
            scale = max(1.0, float(b.abs().max()))
            assert float((a - b.detach()).abs().max()) <= 1e-3 * scale


def test_static_postprocess_equals_reference_path(dev):
    """decode -> multiclass_rnms -> rbbox2result: the sync-free fixed-shape pipeline must return exactly what the
    reference-shaped dynamic path returns (same detections, same per-class order), incl. the > max_per_img re-sort,
    the empty case and the capacity-overflow fallback."""
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.core import (multiclass_rnms, multiclass_rnms_static, rbbox2result,
                                                         rbbox2result_packed)
    rng = np.random.RandomState(3)
    for n, thr, max_num, cap in ((1500, 0.05, 2000, 16384), (1500, 0.05, 300, 16384), (400, 0.999, 2000, 16384),
                                 (1500, 0.05, 2000, 512)):
        d = S.gen_polys(n, 21, clustered=True)
        boxes = _t(d[:, :8], dev)
        scores = np.zeros((n, 16), np.float32)
        scores[:, 1:] = rng.uniform(0, 1, (n, 15)) * (rng.uniform(0, 1, (n, 15)) < 0.08)
        scores = _t(scores, dev)
        rep = _t(rng.uniform(0, 1024, (n, 18)), dev)
        nms_cfg = ConfigDict(type='rnms', iou_thr=0.4)
        want_b, want_l = multiclass_rnms(boxes, scores, thr, nms_cfg, max_num, multi_reppoints=rep)
        want = rbbox2result(want_b, want_l, 16)
        got = rbbox2result_packed(multiclass_rnms_static(boxes, scores, thr, nms_cfg, max_num, rep, capacity=cap), 16)
        if cap < int((scores[:, 1:] > thr).sum()):
            assert got is None
            continue
        assert len(got) == len(want) == 15
        for g_, w_ in zip(got, want):
            if want_b.size(0) > max_num and w_.shape[0] > 0:      # unstable score sort in the reference: compare as sets
                assert sorted(map(tuple, g_.tolist())) == sorted(map(tuple, w_.tolist()))
            else:
                assert np.array_equal(g_, w_)


def test_inference_device_part_is_hipgraph_capturable(dev):
    """backbone -> FPN -> dense head -> decode -> multiclass rotated NMS -> packing as ONE hipGraph: no host sync,
    no allocation through the C ABI, everything on the capture stream.  Replay must reproduce the eager result."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    from orientedreppoints_amd.mmdet_models.core import rbbox2result_packed
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.3)          # logit(0.05) = -2.94: a minority of the pairs pass score_thr
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    img = torch.randn(1, 3, 256, 256, device=dev)
    metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)]

    def device_part():
        outs = head(model.extract_feat(img))
        return head.get_bboxes(*(tuple(outs) + (metas, model.test_cfg, False)), static=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            eager = device_part()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    want = rbbox2result_packed(eager[0], head.num_classes)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        packed = device_part()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    got = rbbox2result_packed(packed[0], head.num_classes)
    assert sum(len(c) for c in want) > 0
    assert [c.shape for c in got] == [c.shape for c in want]
    for a, b in zip(got, want):
        assert np.allclose(a, b, rtol=1e-4, atol=1e-2)


def test_graphed_inference_equals_simple_test(dev):
    """mmdet_models.GraphedInference (the deployment path bench.py times): one hipGraph replay per call must return what
    simple_test_batch returns -- also for a NEW image of the captured shape, and for two images per call."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, GraphedInference, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.3)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    for B in (1, 2):
        metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)] * B
        gi = GraphedInference(model, torch.randn(B, 3, 256, 256, device=dev), metas)
        for seed in (1, 2):
            img = torch.randn(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
            got = gi(img)
            with torch.no_grad():
                want = model.simple_test_batch(img, metas)
            assert len(got) == len(want) == B and sum(len(c) for r in want for c in r) > 0
            for gr, wr in zip(got, want):
                assert [c.shape for c in gr] == [c.shape for c in wr]
                for a, b in zip(gr, wr):
                    assert np.allclose(a, b, rtol=1e-4, atol=1e-2)


def test_aug_test_is_one_nms_over_the_union_of_the_views(dev):
    """Detector.aug_test (reference orientedreppoints_detector.py:111-144): (a) two copies of the same view suppress each
    other -> exactly simple_test's boxes; (b) original + mirrored + half-size views: the result equals ONE multiclass
    rotated NMS over the mapped-back candidates, the mapping restated here in numpy; (c) rescale=False multiplies by the
    first view's scale factor."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    from orientedreppoints_amd.mmdet_models.core import multiclass_rnms, rbbox2result
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.3)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    img = torch.randn(1, 3, 256, 256, device=dev)
    meta = dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)
    with torch.no_grad():
        single = model.simple_test(img, [meta], rescale=True)
        twice = model(img=[img, img], img_meta=[[meta], [meta]], return_loss=False, rescale=True)
    assert sum(len(c) for c in single) > 20
    for a, b in zip(twice, single):
        assert a.shape[1] == 9 and a.shape[0] == b.shape[0]
        assert np.allclose(a, b[:, -9:], rtol=1e-5, atol=1e-4)   # simple_test rows are [reppoints | corners | score]
    views = [(img, meta),
             (img.flip(-1), dict(meta, flip=True)),
             (torch.nn.functional.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=False),
              dict(img_shape=(128, 128, 3), pad_shape=(128, 128, 3), scale_factor=0.5, flip=False))]
    imgs, metas = [v[0] for v in views], [[v[1]] for v in views]
    # aug_test and the restatement below each run the three views' forwards: a candidate sitting on the score threshold or on
    # the NMS threshold must get the same bits both times, so the library's convolutions run in their deterministic mode here
    # (its default picks for the half-size view's tiny maps accumulate with atomics; seen once as 58 vs 57 boxes in a class)
    det_flag = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    with torch.no_grad():
        got = model.aug_test(imgs, metas, rescale=True)
        cand = []
        for im, m in zip(imgs, metas):
            outs = head(model.extract_feat(im))
            b, sc = head.get_bboxes(*(tuple(outs) + (m, model.test_cfg, False, False)))[0]
            b = b.cpu().numpy()
            if m[0]['flip']:
                b = b.copy()
                b[:, 0::2] = np.float32(m[0]['img_shape'][1]) - b[:, 0::2] - np.float32(1)
            cand.append((b / np.float32(m[0]['scale_factor']), sc.cpu().numpy()))
        boxes = torch.from_numpy(np.concatenate([c[0] for c in cand])).to(dev)
        scores = torch.from_numpy(np.concatenate([c[1] for c in cand])).to(dev)
        det, lab = multiclass_rnms(boxes, scores, model.test_cfg.score_thr, model.test_cfg.nms, model.test_cfg.max_per_img)
        want = rbbox2result(det, lab, head.num_classes)
        half_first = model.aug_test(imgs[::-1], metas[::-1], rescale=False)
        full_first = model.aug_test(imgs[::-1], metas[::-1], rescale=True)
    torch.backends.cudnn.deterministic = det_flag
    assert sum(len(c) for c in want) > sum(len(c) for c in single)        # the extra views really add detections
    # (the library's convolution kernels for the 128^2 view's tiny maps are not bitwise reproducible: two forwards of the
    # same view agree to an ulp or two, hence allclose and not array_equal)
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-5, atol=1e-4)
    for a, b in zip(half_first, full_first):
        assert a.shape == b.shape and np.allclose(a[:, :8], b[:, :8] * np.float32(0.5), rtol=1e-5, atol=1e-4)
        assert np.allclose(a[:, 8], b[:, 8], rtol=1e-5, atol=1e-6)


def test_pipelined_inference_returns_each_images_own_results_in_order(dev):
    """mmdet_models.PipelinedInference (bench.py's throughput mode: several captured graphs in flight on their own streams,
    results fetched asynchronously): a stream of DIFFERENT images must come back in order, each with exactly what
    GraphedInference / simple_test_batch return for that image."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, PipelinedInference, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.3)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)]
    imgs = [torch.randn(1, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(10 + i)) for i in range(7)]
    # every image is inferred twice (eagerly and through a graph) and the two must agree row by row: the library's
    # convolutions run in their deterministic mode here (its default picks for these tiny maps accumulate with atomics; a
    # candidate on the score or NMS threshold then differs between two runs of the SAME path -- seen once in four suite runs)
    det_flag = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        _pipelined_body(model, imgs, metas, PipelinedInference)
    finally:
        torch.backends.cudnn.deterministic = det_flag


def _pipelined_body(model, imgs, metas, PipelinedInference):
    with torch.no_grad():
        want = [model.simple_test_batch(im, metas) for im in imgs]
    assert len({sum(len(c) for r in w for c in r) for w in want}) > 1          # the images really differ
    for depth in (2, 3):
        pi = PipelinedInference(model, imgs[0], metas, depth=depth)
        got = []
        for i, im in enumerate(imgs):
            r = pi.submit(im)
            assert (r is None) == (i < depth)
            if r is not None:
                got.append(r)
        got += pi.flush()
        assert len(got) == len(imgs) and pi.flush() == []
        for g, w in zip(got, want):
            for gr, wr in zip(g, w):
                assert [c.shape for c in gr] == [c.shape for c in wr]
                for a, b in zip(gr, wr):
                    assert np.allclose(a, b, rtol=1e-4, atol=1e-2)
    # more (point, class) pairs above score_thr than the static capacity holds: every image falls back to the
    # reference-shaped path and still returns its own results; two images per call
    model.test_cfg['static_capacity'] = 64
    try:
        pi = PipelinedInference(model, imgs[0], metas, depth=2)
        got = [r for r in (pi.submit(im) for im in imgs[:4]) if r is not None] + pi.flush()
        for g, w in zip(got, want[:4]):
            for gr, wr in zip(g, w):
                assert [c.shape for c in gr] == [c.shape for c in wr]
    finally:
        model.test_cfg['static_capacity'] = 8192
    metas2 = metas * 2
    pair = [torch.cat([imgs[0], imgs[1]]), torch.cat([imgs[2], imgs[3]]), torch.cat([imgs[4], imgs[0]])]
    with torch.no_grad():                                   # (the library picks other algorithms at N = 2: compare like with like)
        want2 = [model.simple_test_batch(p, metas2) for p in pair]
    pi = PipelinedInference(model, pair[0], metas2, depth=2)
    got = [r for r in (pi.submit(p) for p in pair) if r is not None] + pi.flush()
    for g, w in zip(got, want2):
        assert len(g) == len(w) == 2
        for gr, wr in zip(g, w):
            assert [c.shape for c in gr] == [c.shape for c in wr]
            for a, b in zip(gr, wr):
                assert np.allclose(a, b, rtol=1e-4, atol=1e-2)


def test_graphed_inference_owns_its_memory_and_follows_weight_updates(dev):
    """A captured graph must survive everything an eager caller does afterwards in the same process: scratch growth (a
    20 k-box fp64 merge NMS, a 16 k-box rnms, the capacity-overflow fallback), cache eviction of the packed weights /
    folded BatchNorm affines, and parameter updates (re-capture)."""
    from orientedreppoints_amd import _lib, synthetic as S
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.dota_devkit.result_merge import py_gpu_nms_poly
    from orientedreppoints_amd.mmdet_models import ConfigDict, GraphedInference, build_detector
    import importlib
    from orientedreppoints_amd.mmdet_ops import rnms
    DC = importlib.import_module('orientedreppoints_amd.mmdet_ops.deform_conv')      # the package re-exports a function of that name
    FN = importlib.import_module('orientedreppoints_amd.mmdet_ops.fused_norm')
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(-3.3)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
    metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)]
    img = torch.randn(1, 3, 256, 256, device=dev)
    _lib._workspaces.clear()                               # start from a small shared scratch
    gi = GraphedInference(model, img, metas)
    with torch.no_grad():
        want = model.simple_test_batch(img, metas)

    def same(got, ref):
        assert sum(len(c) for r in ref for c in r) > 0
        for gr, wr in zip(got, ref):
            assert [c.shape for c in gr] == [c.shape for c in wr]
            for a, b in zip(gr, wr):
                assert np.allclose(a, b, rtol=1e-4, atol=1e-2)
    same(gi(img), want)
    # ---- eager work that grows / replaces every shared buffer ---------------------------------------------------------
    before = {k: v.data_ptr() for k, v in _lib._workspaces.items()}
    py_gpu_nms_poly(S.gen_polys(20000, 3), 0.3)                                        # 20 k-box merge NMS (fp64)
    rnms(torch.from_numpy(S.gen_polys(16000, 4, clustered=True).astype(np.float32)).to(dev), 0.4)
    cur = torch.cuda.current_stream(dev).cuda_stream
    grown = [k for k, v in _lib._workspaces.items() if k[1] == cur and before.get(k) != v.data_ptr()]
    assert grown or not before, "the eager scratch was expected to be replaced by a larger one"
    DC.invalidate_packed_weights()                                                       # cache eviction
    junk = [torch.full((1 << 22,), float('nan'), device=dev) for _ in range(8)]        # recycle freed blocks with NaNs
    del junk
    torch.cuda.synchronize()
    same(gi(img), want)                                    # the replay must not have touched freed memory
    # ---- capacity-overflow fallback inside __call__ (eager simple_test_batch) then a replay again --------------------
    old_bias = head.reppoints_cls_out.bias.detach().clone()
    with torch.no_grad():
        head.reppoints_cls_out.bias.fill_(3.0)             # every (point, class) pair passes score_thr -> overflow
    n0 = gi.captures
    res_over = gi(img)                                     # re-captures (weights changed), overflows, falls back
    assert gi.captures == n0 + 1
    with torch.no_grad():
        same(res_over, model.simple_test_batch(img, metas))
        head.reppoints_cls_out.bias.copy_(old_bias)
    got = gi(img)                                          # weights changed back -> another capture, then parity
    assert gi.captures == n0 + 2
    same(got, want)
    # an optimizer-style in-place update of a DeformConv weight must reach the graph as well
    with torch.no_grad():
        head.reppoints_cls_conv.weight.mul_(1.5)
        want2 = model.simple_test_batch(img, metas)
    same(gi(img), want2)


@pytest.mark.parametrize("size,max_per_img,thr_bias", [(256, 2000, -3.3), (384, 150, -3.0), (256, 2000, -9.0)])
def test_fused_postprocess_equals_tensor_op_path(dev, size, max_per_img, thr_bias):
    """csrc/orp_postproc.hip (gather / compaction / packing kernels around min-area-rect and the NMS) against the
    tensor-op static path on the same head outputs: identical detections in identical order -- incl. per-level top-k
    (nms_pre smaller than a level), the > max_per_img re-sort and the empty case."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    from orientedreppoints_amd.mmdet_models.core import rbbox2result_packed
    torch.manual_seed(0)
    cfg = dict(test_cfg)
    cfg.update(nms_pre=300, max_per_img=max_per_img)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(cfg)).to(dev).eval()
    head = model.bbox_head
    with torch.no_grad():
        head.reppoints_cls_out.weight.normal_(0, 0.05)
        head.reppoints_cls_out.bias.fill_(thr_bias)
        head.reppoints_pts_init_out.bias.copy_(torch.tensor(
            [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
            dtype=torch.float32, device=dev).reshape(-1) * 2.0)
        head.reppoints_pts_refine_out.weight.normal_(0, 0.05)
    img = torch.randn(1, 3, size, size, device=dev)
    metas = [dict(img_shape=(size, size, 3), pad_shape=(size, size, 3), scale_factor=1.0, flip=False)]
    with torch.no_grad():
        outs = head(model.extract_feat(img))
        res = {}
        for fused in (True, False):
            model.test_cfg['fused_postprocess'] = fused
            packed = head.get_bboxes(*(tuple(outs) + (metas, model.test_cfg, False)), static=True)[0]
            res[fused] = rbbox2result_packed(packed, head.num_classes)
    n_det = sum(len(c) for c in res[False])
    if thr_bias < -8:
        assert n_det == 0
    else:
        assert n_det > 50
    for a, b in zip(res[True], res[False]):
        if max_per_img < 2000 and len(b):                # unstable score order on the re-sort branch: compare as sets
            assert sorted(map(tuple, a.tolist())) == sorted(map(tuple, b.tolist()))
        else:
            assert np.array_equal(a, b)
