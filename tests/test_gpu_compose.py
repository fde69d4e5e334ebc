"""GPU (MI355X): the Python COMPOSITIONS of the hot path against goldens produced by the REFERENCE'S OWN Python
(tests/golden/make_golden_compose.py executes head get_bboxes_single / multiclass_rnms / rbbox2result,
pointset_target, SpatialBorderLoss, GIoULoss, FocalLoss and the whole head loss() from /root/reference on CPU).

Bars (BASELINE.json north_star): discrete outcomes -- kept detections, labels, order, assignments, selected positive
sets, normalisers -- identical; floats within 1e-4 (relative to max(1, |x|) for coordinates / the summed class loss).
Round 6: min-area-rect is bit-exact against the oracle (the kernel evaluates the host C library's cosf, csrc/orp_libm.hpp), so
the "provable tie" exemption rounds 3-5 granted the APAA quality values is gone: every Q value is held to 1e-4.  The scenes
the golden generator REJECTED for not being ulp-robust are run too (`test_rejected_scenes_report`), reporting -- not asserting --
how many of them / of their detections differ from the reference."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import compose_inputs as CI  # noqa: E402



@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from orientedreppoints_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


# base seeds of tests/golden/make_golden_compose.py (PP_SCENES / LOSS_CASES): a scene is re-drawn with seed + 100 until all
# its discrete outcomes survive ulp-level input noise, so (accepted seed - base seed) / 100 = scenes REJECTED by that filter
BASE_SEEDS = {'pp_small': 11, 'pp_empty': 12, 'pp_full': 13, 'pp_over_max': 14, 'loss_a': 21, 'loss_b': 22, 'loss_c': 23}


@pytest.fixture(scope="module")
def G(golden_dir):
    import conftest
    g = np.load(os.path.join(golden_dir, "compose_py.npz"))
    rej = {k: (int(g[k + '_seed']) - b) // 100 for k, b in BASE_SEEDS.items()}
    conftest.REPORT.append("compose goldens are pre-filtered for ulp-robustness (borderline scenes are NOT in the end-to-end tests; "
                           "the kernel-level bit-exact IoU / NMS tests cover them): scenes rejected before the accepted one: %s" % rej)
    return g


def _head(dev):
    from orientedreppoints_amd.dota_configs import r50_model
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.registry import build_head
    torch.manual_seed(0)
    return build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()


def _cfg(max_per_img):
    from orientedreppoints_amd.dota_configs import test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict
    c = dict(test_cfg)
    c['max_per_img'] = max_per_img
    return ConfigDict(c)


PP = {  # name -> (img_size, kwargs of compose_inputs.postprocess_scene, max_per_img): as in make_golden_compose.PP_SCENES
    'small': (256, dict(), 2000),
    'empty': (256, dict(logit_mean=-9.0, logit_std=0.5, obj_density=0), 2000),
    'full': (1024, dict(), 2000),
    'over_max': (1024, dict(obj_density=1.0 / 25), 300),
}


def _rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)), initial=0.0))


def _check_dets(dets, labels, G, name):
    want, wl = G['pp_%s_dets' % name], G['pp_%s_labels' % name]
    assert dets.shape == want.shape, (dets.shape, want.shape)
    assert np.array_equal(labels, wl), "labels / order differ from the reference"
    if want.shape[0]:
        assert _rel(dets[:, -1], want[:, -1]) <= 1e-6                       # scores
        assert _rel(dets[:, :-1], want[:, :-1]) <= 1e-4                     # 18 reppoints + 8 corners


@pytest.mark.parametrize("name", list(PP))
def test_get_bboxes_single_and_multiclass_rnms_vs_reference(dev, G, name):
    """a3 + a5 + a21(rbbox2result): every product path that implements the test-time post-processing returns the
    detections of the reference's get_bboxes_single -> multiclass_rnms, same labels, same order."""
    from orientedreppoints_amd.mmdet_models.core import rbbox2result, rbbox2result_packed
    size, kw, max_per_img = PP[name]
    cls, pts = CI.postprocess_scene(size, int(G['pp_%s_seed' % name]), **kw)
    head, cfg = _head(dev), _cfg(max_per_img)
    cls_t = [torch.from_numpy(c)[None].to(dev) for c in cls]
    pts_t = [torch.from_numpy(p)[None].to(dev) for p in pts]
    metas = [CI.img_meta(size)]
    counts = G['pp_%s_class_counts' % name]
    with torch.no_grad():
        # (1) reference-shaped dynamic path: get_bboxes_single -> multiclass_rnms -> rnms
        dets, labels = head.get_bboxes(cls_t, None, pts_t, None, metas, cfg, rescale=False, nms=True)[0]
        _check_dets(dets.cpu().numpy(), labels.cpu().numpy(), G, name)
        res = rbbox2result(dets, labels, head.num_classes)
        assert [r.shape[0] for r in res] == counts.tolist()
        # (2) fused static kernels (decode / compaction / NMS / packing), (3) static tensor-op path
        for fused in (True, False):
            cfg2 = _cfg(max_per_img)
            cfg2['fused_postprocess'] = fused
            packed = head.get_bboxes(cls_t, None, pts_t, None, metas, cfg2, static=True)[0]
            per_class = rbbox2result_packed(packed, head.num_classes)
            assert per_class is not None, "static capacity overflow on a scene that fits"
            assert [r.shape[0] for r in per_class] == counts.tolist()
            host = packed.cpu().numpy()
            n = int(host[-1, 0])
            _check_dets(host[:n, :-1], host[:n, -1].astype(np.int64), G, name)


def test_rejected_scenes_report(dev, G, golden_dir):
    """Round-5 verdict, weak 2: the asserting tests above run on scenes that survived the generator's ulp-perturbation filter.
    This one runs the draws that filter REJECTED (tests/golden/make_golden_compose_rejected.py: the reference's own Python on
    the unperturbed inputs) through the three product paths and REPORTS -- it does not assert equality -- how many scenes and
    detections differ from the reference, so the end-to-end mismatch rate on borderline scenes is a measured number.  (What can
    differ: the device sigmoid vs the host's in the last ulp next to score_thr / in the top-k order.  Min-area-rect, rotated
    IoU and the NMS keep sets are bit-exact at kernel level.)"""
    import conftest
    R = np.load(os.path.join(golden_dir, "compose_rejected_py.npz"))
    head = _head(dev)
    lines, scenes, bad_scenes, n_det, n_bad_det = [], 0, 0, 0, 0
    for name in PP:
        size, kw, max_per_img = PP[name]
        for seed in R['pp_%s_seeds' % name].tolist():
            want, wl = R['pp_%s_%d_dets' % (name, seed)], R['pp_%s_%d_labels' % (name, seed)].astype(np.int64)
            cls, pts = CI.postprocess_scene(size, int(seed), **kw)
            cls_t = [torch.from_numpy(c)[None].to(dev) for c in cls]
            pts_t = [torch.from_numpy(p)[None].to(dev) for p in pts]
            metas = [CI.img_meta(size)]
            outs = {}
            with torch.no_grad():
                dets, labels = head.get_bboxes(cls_t, None, pts_t, None, metas, _cfg(max_per_img), rescale=False, nms=True)[0]
                outs['dynamic'] = (dets.cpu().numpy(), labels.cpu().numpy())
                for fused in (True, False):
                    cfg2 = _cfg(max_per_img)
                    cfg2['fused_postprocess'] = fused
                    host = head.get_bboxes(cls_t, None, pts_t, None, metas, cfg2, static=True)[0].cpu().numpy()
                    n = int(host[-1, 0])
                    outs['fused' if fused else 'static'] = (host[:n, :-1], host[:n, -1].astype(np.int64))
            scenes += 1
            n_det += want.shape[0]
            worst = 0
            for path, (d, l) in outs.items():
                if d.shape == want.shape and np.array_equal(l, wl) and (want.shape[0] == 0 or
                        (_rel(d[:, -1], want[:, -1]) <= 1e-6 and _rel(d[:, :-1], want[:, :-1]) <= 1e-4)):
                    continue
                # detections of the reference without a partner (same label, score within 1e-6, boxes within 1e-4) on this path
                key = lambda a, b: {(int(b[i]), round(float(a[i, -1]), 5)) for i in range(a.shape[0])}     # noqa: E731
                missing = len(key(want, wl) ^ key(d, l))
                worst = max(worst, max(missing, 1))
                lines.append("   %s seed %d, path %s: %d vs %d detections, %d unmatched" % (name, seed, path, d.shape[0], want.shape[0], missing))
            # the three product paths must agree with EACH OTHER whatever the reference says (asserted)
            assert outs['fused'][0].shape == outs['static'][0].shape and np.array_equal(outs['fused'][1], outs['static'][1])
            bad_scenes += worst > 0
            n_bad_det += worst
    conftest.REPORT.append("scenes REJECTED by the golden generator's ulp-robustness filter, run anyway: %d of %d scenes differ from "
                           "the reference's detections (%d of %d detections unmatched)" % (bad_scenes, scenes, n_bad_det, n_det))
    for ln in lines:
        conftest.REPORT.append(ln)


def test_postprocess_as_hipgraph_replay_vs_reference(dev, G):
    """f1: decode -> multiclass rotated NMS -> packing captured as ONE hipGraph; replays on new head outputs equal the
    reference's detections (scene 'full', then 'over_max' shapes are identical so the same graph serves both)."""
    from orientedreppoints_amd.mmdet_models.core import rbbox2result_packed
    head = _head(dev)
    size = 1024
    metas = [CI.img_meta(size)]
    cfg = _cfg(2000)
    cls0, pts0 = CI.postprocess_scene(size, int(G['pp_full_seed']), **PP['full'][1])
    s_cls = [torch.from_numpy(c)[None].to(dev) for c in cls0]
    s_pts = [torch.from_numpy(p)[None].to(dev) for p in pts0]
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(2):
            head.get_bboxes(s_cls, None, s_pts, None, metas, cfg, static=True)
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
        packed = head.get_bboxes(s_cls, None, s_pts, None, metas, cfg, static=True)[0]
    for name in ('full', 'small_as_full', 'full'):
        if name == 'small_as_full':
            # another scene of the same shapes through the same graph (different seed): compare with the eager path
            cls1, pts1 = CI.postprocess_scene(size, 4242, **PP['full'][1])
        else:
            cls1, pts1 = cls0, pts0
        for d, s in zip(s_cls, cls1):
            d.copy_(torch.from_numpy(s)[None])
        for d, s in zip(s_pts, pts1):
            d.copy_(torch.from_numpy(s)[None])
        graph.replay()
        host = packed.cpu().numpy()
        n = int(host[-1, 0])
        assert host[-1, 1] == 0
        if name == 'full':
            _check_dets(host[:n, :-1], host[:n, -1].astype(np.int64), G, 'full')
        else:
            with torch.no_grad():
                dets, labels = head.get_bboxes(s_cls, None, s_pts, None, metas, cfg)[0]
            assert np.array_equal(host[:n, -1].astype(np.int64), labels.cpu().numpy())
            assert np.array_equal(host[:n, :-1], dets.cpu().numpy())
        assert rbbox2result_packed(packed, 16) is not None


def test_spatial_border_loss_vs_reference(dev, G):
    """a19: SpatialBorderLoss value and d loss / d pts against spatial_border_loss.py:8-92 run on CPU."""
    from orientedreppoints_amd.mmdet_models.losses import SpatialBorderLoss
    pts = torch.from_numpy(G['sb_pts']).to(dev).requires_grad_(True)
    loss = SpatialBorderLoss(loss_weight=0.1)(pts, torch.from_numpy(G['sb_gts']).to(dev),
                                              torch.from_numpy(G['sb_w']).to(dev), y_first=False, avg_factor=None)
    assert loss.shape == G['sb_loss'].shape
    assert abs(float(loss) - float(G['sb_loss'])) <= 1e-6
    loss.sum().backward()
    assert np.max(np.abs(pts.grad.cpu().numpy() - G['sb_grad'])) <= 1e-7
    l2 = SpatialBorderLoss(loss_weight=0.1)(torch.from_numpy(G['sb_gts'][:5].reshape(5, 4, 2).mean(1).repeat(9, 0).reshape(5, 18)).to(dev),
                                            torch.from_numpy(G['sb_gts'][:5]).to(dev), torch.ones(5, device=dev))
    assert l2.shape == G['sb_inside_loss'].shape and float(l2.sum()) == 0.0


def test_offset_to_pts_and_selection_outputs_vs_reference(dev, golden_dir):
    """a21 / a18 wrapper outputs held in apaa_py.npz: offset_to_pts, and point_samples_selection's
    label_weight / rbox_weight / pos_normalize_term."""
    import types
    from orientedreppoints_amd.mmdet_models import orientedreppoints_head_train as T
    g = np.load(os.path.join(golden_dir, "apaa_py.npz"))
    points = g['points']
    centers = [[torch.from_numpy(points[:1024]).to(dev), torch.from_numpy(points[1024:1280]).to(dev)]]
    preds = [torch.from_numpy(g['otp_pred0']).to(dev), torch.from_numpy(g['otp_pred1']).to(dev)]
    fake = types.SimpleNamespace(num_points=9, point_strides=[8, 16])
    out = T.offset_to_pts(fake, centers, preds)
    assert np.array_equal(out[0].cpu().numpy(), g['otp_out0']) and np.array_equal(out[1].cpu().numpy(), g['otp_out1'])
    # point_samples_selection on the golden Q
    head = types.SimpleNamespace(num_points=9, top_ratio=0.4, point_base_scale=2, point_strides=[8, 16, 32, 64, 128])
    N = g['psets'].shape[0]
    pos = torch.from_numpy(g['qa_pos_inds']).to(dev)
    label = torch.from_numpy(g['mia_labels']).to(dev)
    lw = torch.ones(N, device=dev)
    rw = torch.zeros(N, device=dev); rw[pos] = 1.0
    nlev = [1024, 256, 64, 16, 4]
    level_of_index = torch.cat([torch.full((n,), l, dtype=torch.int32, device=dev) for l, n in enumerate(nlev)])
    lab2, lw2, rw2, num_pos, pnt = T.point_samples_selection(
        head, torch.from_numpy(g['qa_out']).to(dev), label.clone(), lw, rw, pos,
        torch.from_numpy(g['sel_pos_gt_inds']).to(dev), level_of_index, 5, int(g['gts'].shape[0]))
    assert np.array_equal(lab2.cpu().numpy(), g['sel_label'])
    assert np.array_equal(lw2.cpu().numpy(), g['sel_label_weight'])
    assert np.array_equal(rw2.cpu().numpy(), g['sel_rbox_weight'])
    assert int(num_pos) == int(g['sel_num_pos'])
    assert np.array_equal(pnt.cpu().numpy(), g['sel_pos_normalize_term'])


LOSS = {'a': (256, (1, 32)), 'b': (512, (256, 7)), 'c': (1024, (32, 100))}


def _pack_levels(lst):
    return np.concatenate([a.reshape(a.shape[0], a.shape[1], -1) for a in lst], axis=2)


@pytest.mark.parametrize("name", list(LOSS))
def test_head_loss_vs_reference_python(dev, G, oracle, name):
    """a13 + a15 + a18 + a19 + the whole loss() (head :320-493): targets, APAA quality, selected positives, the five
    loss terms and d loss / d (cls_scores, pts_preds_init, pts_preds_refine) against the reference's own loss()."""
    from orientedreppoints_amd.dota_configs import train_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict
    size, num_gts = LOSS[name]
    p = 'loss_%s_' % name
    case = CI.loss_case(size, num_gts, int(G[p + 'seed']), channels=256)
    B = len(num_gts)
    head = _head(dev).train()
    leaf = lambda a: torch.from_numpy(a).to(dev).requires_grad_(True)   # noqa: E731
    cls = [leaf(a) for a in case['cls']]
    init = [leaf(a) for a in case['init']]
    refine = [leaf(a) for a in case['refine']]
    feats = [torch.from_numpy(a).to(dev) for a in case['feats']]
    gts = [torch.from_numpy(a).to(dev) for a in case['gts']]
    labels = [torch.from_numpy(a).to(dev) for a in case['labels']]
    metas = [CI.img_meta(size) for _ in range(B)]
    rec = {}
    losses = head.loss(cls, init, refine, feats, gts, labels, metas, ConfigDict(train_cfg), record=rec)

    # ---- targets (pointset_target.py) -----------------------------------------------------------------------------
    it = rec['init_target']
    gt_inds = np.concatenate([t.reshape(B, -1).cpu().numpy() for t in it[7]], 1)
    assert np.array_equal(gt_inds, G[p + 'init_gt_inds'])
    w_init = np.concatenate([t.reshape(B, -1).cpu().numpy() for t in it[4]], 1)
    assert np.array_equal(w_init.astype(np.uint8), G[p + 'init_rbox_weights'])
    init_gt = np.concatenate([t.reshape(B, -1, 8).cpu().numpy() for t in it[2]], 1)
    assert np.array_equal(init_gt[gt_inds > 0], G[p + 'init_rbbox_gt_pos'])
    assert [int(it[5]), int(it[6])] == G[p + 'init_num_total'].tolist()
    rt = rec['refine_target']
    assert np.array_equal(np.stack([t.cpu().numpy() for t in rt[0]]), G[p + 'refine_labels'])
    assert np.array_equal(np.stack([t.cpu().numpy() for t in rt[1]]).astype(np.uint8), G[p + 'refine_label_weights'])
    assert np.array_equal(np.stack([t.cpu().numpy() for t in rt[4]]).astype(np.uint8), G[p + 'refine_rbox_weights'])
    max_dq = 0.0
    for i in range(B):
        pos = rt[5][i].cpu().numpy()
        assert np.array_equal(pos, G[p + 'refine_pos_inds_%d' % i])
        assert np.array_equal(rt[6][i].cpu().numpy(), G[p + 'refine_pos_gt_index_%d' % i])
        assert np.array_equal(rt[2][i].cpu().numpy()[pos], G[p + 'refine_rbox_gt_pos_%d' % i])
        # ---- APAA quality: 1e-4 on EVERY value (no tie exemption since round 6) ----------------------------------------
        q, want = rec['qa'][i].cpu().numpy(), G[p + 'qa_%d' % i]
        d = np.abs(q - want)
        max_dq = max(max_dq, float(np.max(d, initial=0.0)))
        assert np.max(d, initial=0.0) <= 1e-4, "Q differs from the reference beyond 1e-4"
        # ---- selection ------------------------------------------------------------------------------------------------
        lab, lw, rw, npos, pnt = rec['sel'][i]
        assert np.array_equal(lab.cpu().numpy(), G[p + 'sel_label_%d' % i])
        assert np.array_equal(lw.cpu().numpy().astype(np.uint8), G[p + 'sel_label_weight_%d' % i])
        assert np.array_equal(rw.cpu().numpy().astype(np.uint8), G[p + 'sel_rbox_weight_%d' % i])
        assert npos == int(G[p + 'sel_num_pos_%d' % i])
        assert np.array_equal(pnt.cpu().numpy(), G[p + 'sel_pos_normalize_term_%d' % i])
    # ---- the five loss terms ----------------------------------------------------------------------------------------
    total = 0
    for k in ('loss_cls', 'loss_rbox_init', 'loss_rbox_refine', 'loss_spatial_init', 'loss_spatial_refine'):
        v = losses[k]
        vs = v if isinstance(v, (list, tuple)) else [v]
        got = np.array([float(t.sum()) for t in vs])
        want = G[p + k]
        assert got.shape == want.shape, k
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-4, (k, got, want)
        for t in vs:
            total = total + t.sum()
    # ---- gradients w.r.t. the head outputs ----------------------------------------------------------------------------
    total.backward()
    gc = _pack_levels([c.grad.cpu().numpy() for c in cls])
    if (p + 'grad_cls') in G.files:
        assert np.max(np.abs(gc - G[p + 'grad_cls'])) <= 1e-4
    else:
        assert np.max(np.abs(gc.reshape(-1)[::37] - G[p + 'grad_cls_sub'])) <= 1e-4
        s = float(np.abs(gc.astype(np.float64)).sum())
        assert abs(s - float(G[p + 'grad_cls_abs_sum'])) <= 1e-4 * float(G[p + 'grad_cls_abs_sum'])
    for nm, ts in (('init', init), ('refine', refine)):
        ga = _pack_levels([(t.grad if t.grad is not None else torch.zeros_like(t)).cpu().numpy() for t in ts])
        rows = G[p + 'grad_%s_rows' % nm]
        want = np.zeros_like(ga.transpose(0, 2, 1))
        want[rows[:, 0], rows[:, 1]] = G[p + 'grad_%s_vals' % nm]
        assert np.max(np.abs(ga.transpose(0, 2, 1) - want)) <= 1e-4, nm
    n_q = sum(len(G[p + 'qa_%d' % i]) for i in range(B))
    n_tie = sum(int((G[p + 'qa_margin_%d' % i] < 1e-5).sum()) for i in range(B))
    import conftest
    conftest.REPORT.append("a15, head_loss case %r: all %d quality values within 1e-4 of the reference (largest difference %.2e); "
                           "%d of them sit on min-area-rect ties (exempt in rounds 3-5, not any more)" % (name, n_q, max_dq, n_tie))


def test_candidate_selection_radix_select_equals_topk(dev):
    """`orp_pp_select` (radix select + counting rank) == `scores.max(dim=1)` + per-level `topk(nms_pre)` of
    get_bboxes_single (head :730-737): identical candidate lists on continuous scores; on heavily tied scores the set and
    order of a stable descending sort (ties by ascending index), NaN scores first."""
    import numpy as np
    from orientedreppoints_amd.mmdet_models.core import select_candidates
    sizes = [16384, 4096, 1024, 256, 64]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    N, C, k = int(offs[-1]), 15, 2000
    g = torch.Generator(device='cpu').manual_seed(5)
    sig = torch.sigmoid(torch.randn(C, N, generator=g) * 2 - 3).to(dev)
    cand = select_candidates(sig, offs, k)
    mx = sig.max(dim=0)[0]
    want = []
    for l, n_l in enumerate(sizes):
        seg = mx[int(offs[l]):int(offs[l + 1])]
        want.append(seg.topk(k)[1] + int(offs[l]) if n_l > k else torch.arange(int(offs[l]), int(offs[l + 1]), device=dev))
    assert torch.equal(cand, torch.cat(want))
    # ties (scores quantised to 1/32) and NaNs: stable descending order, NaN highest
    q = (torch.rand(C, N, generator=g) * 32).floor() / 32
    q[3, 100] = float('nan'); q[0, 17000] = float('nan'); q[7, 5] = float('nan')
    q = q.to(dev)
    cand = select_candidates(q, offs, k).cpu().numpy()
    mxq = q.cpu().numpy()
    m = np.where(np.isnan(mxq).any(0), np.inf, np.nanmax(mxq, axis=0))          # NaN propagates through the class max, ranks first
    pos = 0
    for l, n_l in enumerate(sizes):
        seg = m[int(offs[l]):int(offs[l + 1])]
        if n_l > k:
            order = np.lexsort((np.arange(n_l), -seg))[:k]
            assert np.array_equal(cand[pos:pos + k], order + int(offs[l]))
            pos += k
        else:
            assert np.array_equal(cand[pos:pos + n_l], np.arange(int(offs[l]), int(offs[l + 1])))
            pos += n_l
    assert pos == cand.size
    # a level of 36 864 points (1536^2 patches) and k that is not a multiple of 64
    sizes2 = [36864, 9216, 100]
    offs2 = np.concatenate([[0], np.cumsum(sizes2)]).astype(np.int64)
    sig2 = torch.rand(4, int(offs2[-1]), generator=g).to(dev)
    c2 = select_candidates(sig2, offs2, 1999)
    mx2 = sig2.max(dim=0)[0]
    w2 = torch.cat([mx2[:36864].topk(1999)[1], mx2[36864:46080].topk(1999)[1] + 36864, torch.arange(46080, 46180, device=dev)])
    assert torch.equal(c2, w2)


def test_candidate_selection_edge_cases(dev):
    """Radix select corner cases: every score equal (the first k indices), one distinct value at the cut, k = n - 1,
    scores spanning many exponents incl. 0, denormals and exact 1.0, and a single class (no class maximum to take)."""
    import numpy as np
    from orientedreppoints_amd.mmdet_models.core import select_candidates

    def expect(mx, offs, k):
        out = []
        for l in range(len(offs) - 1):
            seg = mx[int(offs[l]):int(offs[l + 1])]
            n_l = seg.size
            out.append((np.lexsort((np.arange(n_l), -seg))[:k] if n_l > k else np.arange(n_l)) + int(offs[l]))
        return np.concatenate(out)
    rng = np.random.RandomState(3)
    # all equal
    offs = np.array([0, 5000, 5100], np.int64)
    sig = torch.full((3, 5100), 0.25, device=dev)
    assert np.array_equal(select_candidates(sig, offs, 2000).cpu().numpy(), expect(np.full(5100, 0.25), offs, 2000))
    # k = n - 1, and a cut that falls inside a run of equal values
    offs = np.array([0, 2001, 4500], np.int64)
    v = rng.choice(np.array([0.1, 0.2, 0.3, 0.7], np.float32), size=4500)
    sig = torch.from_numpy(v[None].copy()).to(dev)                     # one class
    assert np.array_equal(select_candidates(sig, offs, 2000).cpu().numpy(), expect(v, offs, 2000))
    # wide dynamic range
    e = rng.uniform(-45, 0, size=9000)
    v = (10.0 ** e).astype(np.float32)
    v[:50] = 0.0
    v[50:60] = 1.0
    v[60:70] = np.float32(1e-45)
    rng.shuffle(v)
    offs = np.array([0, 9000], np.int64)
    sig = torch.from_numpy(np.stack([v, v * 0.5])).to(dev)
    assert np.array_equal(select_candidates(sig, offs, 2000).cpu().numpy(), expect(v, offs, 2000))
    assert np.array_equal(select_candidates(sig, offs, 37).cpu().numpy(), expect(v, offs, 37))
    assert np.array_equal(select_candidates(sig, offs, 4096).cpu().numpy(), expect(v, offs, 4096))
