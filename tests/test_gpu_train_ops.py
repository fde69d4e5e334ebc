"""GPU: the batched glue operators of the training path (csrc/orp_train.hip, mmdet_ops/train_ops.py) against plain tensor
formulations of what the reference does per image / per level (pointset_target.py:61-121, head :204-222, :250-292, :378-381)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from orientedreppoints_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _levels(dev, B, C, sizes, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [torch.randn(B, C, h, w, generator=g).to(dev) for h, w in sizes]


def test_pointset_target_equals_per_image_reference_semantics(dev):
    from orientedreppoints_amd.mmdet_ops import train_ops
    rng = np.random.RandomState(0)
    B, N, D = 3, 1000, 18
    ks = [5, 0, 17]
    gt_boxes = torch.from_numpy(rng.uniform(0, 500, size=(sum(ks), 8)).astype(np.float32)).to(dev)
    gt_labels = torch.from_numpy(rng.randint(1, 16, size=sum(ks))).to(dev)
    offs = torch.tensor(np.concatenate([[0], np.cumsum(ks)]), dtype=torch.int32, device=dev)
    gi = np.zeros((B, N), np.int64)
    for b, k in enumerate(ks):
        gi[b] = rng.randint(-1, k + 1, size=N) if k else rng.randint(-1, 1, size=N)
    gt_inds = torch.from_numpy(gi).to(dev)
    valid = torch.from_numpy(rng.rand(B, N) > 0.2).to(dev)
    props = torch.from_numpy(rng.normal(size=(B, N, D)).astype(np.float32)).to(dev)
    for use_valid, use_labels, pos_w in ((True, True, -1.0), (False, True, 2.5), (True, False, -1.0)):
        t = train_ops.pointset_target(gt_inds, valid if use_valid else None, gt_boxes, gt_labels if use_labels else None, offs,
                                      pos_weight=pos_w, proposals=props)
        for b, k in enumerate(ks):
            v = valid[b].cpu().numpy() if use_valid else np.ones(N, bool)
            g = np.where(v, gi[b], 0)                                         # unmap(fill = 0)
            pos, neg = v & (g > 0), v & (g == 0)
            lab = np.zeros(N, np.int64)
            box = np.zeros((N, 8), np.float32)
            if k:
                gl = gt_labels.cpu().numpy()[int(offs[b]):int(offs[b + 1])]
                gb = gt_boxes.cpu().numpy()[int(offs[b]):int(offs[b + 1])]
                lab[pos] = gl[g[pos] - 1] if use_labels else 1
                box[pos] = gb[g[pos] - 1]
            assert np.array_equal(t['labels'][b].cpu().numpy(), lab)
            lw = np.where(pos, 1.0 if pos_w <= 0 else pos_w, np.where(neg, 1.0, 0.0)).astype(np.float32)
            assert np.array_equal(t['label_weights'][b].cpu().numpy(), lw)
            assert np.array_equal(t['rbbox_gt'][b].cpu().numpy(), box)
            assert np.array_equal(t['proposal_weights'][b].cpu().numpy(), pos.astype(np.float32))
            assert np.array_equal(t['gt_inds'][b].cpu().numpy(), g)
            assert np.array_equal(t['pos_proposals'][b].cpu().numpy(), np.where(pos[:, None], props[b].cpu().numpy(), 0))
            assert t['counts'][b].tolist() == [int(pos.sum()), int(neg.sum())]


def test_points_from_offsets_and_gather_levels_vs_tensor_ops(dev):
    from orientedreppoints_amd.mmdet_ops import train_ops
    sizes, strides = [(9, 13), (5, 7), (2, 3)], [8, 16, 32]
    B = 2
    lv = [t.requires_grad_(True) for t in _levels(dev, B, 18, sizes, 1)]
    N = sum(h * w for h, w in sizes)

    def centres(h, w, s):
        ys, xs = torch.meshgrid(torch.arange(h, device=dev) * float(s), torch.arange(w, device=dev) * float(s), indexing="ij")
        return torch.stack([xs.reshape(-1), ys.reshape(-1)], 1)
    # offset_to_pts (head :204-222) and the un-swapped refine proposals (:378-381)
    want0, want1 = [], []
    for t, (h, w), s in zip(lv, sizes, strides):
        yx = t.permute(0, 2, 3, 1).reshape(B, -1, 9, 2)
        c = centres(h, w, s)
        want0.append(yx.flip(-1).reshape(B, -1, 18) * s + c.repeat(1, 9))
        want1.append(t.permute(0, 2, 3, 1).reshape(B, -1, 18) * s + c.repeat(1, 9))
    want0, want1 = torch.cat(want0, 1), torch.cat(want1, 1)
    assert torch.equal(train_ops.points_from_offsets(lv, strides, 0), want0.detach())
    assert torch.equal(train_ops.points_from_offsets(lv, strides, 1), want1.detach())
    # gather: forward values and the gradient routed back into the level tensors
    idx = torch.from_numpy(np.random.RandomState(2).choice(B * N, size=37, replace=False)).to(dev).sort()[0]
    got = train_ops.gather_levels(lv, strides, idx, mode=1)
    assert torch.equal(got, want0.reshape(-1, 18)[idx].detach())
    wgt = torch.randn(37, 18, device=dev)
    (got * wgt).sum().backward()
    g_mine = [t.grad.clone() for t in lv]
    for t in lv:
        t.grad = None
    (want0.reshape(-1, 18)[idx] * wgt).sum().backward()
    for a, t in zip(g_mine, lv):
        assert torch.allclose(a, t.grad, rtol=0, atol=1e-6)
    cls = _levels(dev, B, 15, sizes, 3)
    flat = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, 15) for c in cls], 1).reshape(-1, 15)
    assert torch.equal(train_ops.gather_levels(cls, strides, idx, mode=0), flat[idx])
    assert train_ops.gather_levels(cls, strides, idx[:0], mode=0).shape == (0, 15)


def test_outline_samples_equals_linspace_formula(dev):
    from orientedreppoints_amd.mmdet_ops import train_ops
    c = torch.from_numpy(np.random.RandomState(4).uniform(0, 800, size=(50, 8)).astype(np.float32)).to(dev)
    q = c.reshape(-1, 4, 2)
    nxt = torch.roll(q, shifts=-1, dims=1)
    r = torch.linspace(0, 1, 10, device=dev).view(1, 1, 10, 1)
    want = (r * nxt.unsqueeze(2) + (1 - r) * q.unsqueeze(2)).reshape(50, 40, 2)
    got = train_ops.outline_samples(c, 10)
    assert float((got - want).abs().max()) <= 1e-4          # the ratios are linspace's own floats; products may round differently by 1 ulp
    assert train_ops.outline_samples(c[:0], 10).shape == (0, 40, 2)


def test_pointset_target_api_wrappers(dev):
    """init_ / refine_pointset_target keep the reference's tuples: level slices for the init stage, per-image lists + positive
    indices for the refine stage, on a batch whose images have different numbers of gts (one of them none)."""
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.dota_configs import train_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.core import PointGenerator
    from orientedreppoints_amd.mmdet_models.pointset_target import init_pointset_target, refine_pointset_target
    cfg = ConfigDict(train_cfg)
    sizes, strides = [(32, 32), (16, 16), (8, 8)], [8, 16, 32]
    pg = PointGenerator()
    pts = [pg.grid_points(s, st, dev) for s, st in zip(sizes, strides)]
    flags = [torch.ones(h * w, dtype=torch.bool, device=dev) for h, w in sizes]
    gts = [torch.from_numpy(S.gen_polys(6, 1, wh=(16, 80))[:, :8].astype(np.float32) / 4).to(dev),
           torch.zeros((0, 8), device=dev)]
    labels = [torch.randint(1, 16, (6,), device=dev), torch.zeros((0,), dtype=torch.long, device=dev)]
    metas = [dict(pad_shape=(256, 256, 3))] * 2
    out = init_pointset_target([list(pts), list(pts)], [list(flags), list(flags)], gts, metas, cfg.init,
                               gt_labels_list=labels, sampling=False)
    assert len(out) == 8 and [t.shape for t in out[0]] == [(2, 1024), (2, 256), (2, 64)]
    gi = torch.cat([t.reshape(2, -1) for t in out[7]], 1)
    assert int((gi[0] > 0).sum()) == 6 and int((gi[1] > 0).sum()) == 0          # pos_num = 1: one point per gt
    assert out[5] == 6 + 1 and out[6] == (1344 - 6) + 1344                      # max(num, 1) per image, summed
    w = torch.cat([t.reshape(2, -1) for t in out[4]], 1)
    assert torch.equal(w > 0, gi > 0)
    boxes = torch.cat([t.reshape(2, -1, 8) for t in out[2]], 1)
    assert torch.equal(boxes[0][gi[0] > 0], gts[0][gi[0][gi[0] > 0] - 1])
    psets = [[(p[:, :2].repeat(1, 9) + torch.randn(p.size(0), 18, device=dev) * 3) for p in pts] for _ in range(2)]
    ref = refine_pointset_target(psets, [list(flags), list(flags)], gts, metas, cfg.refine, gt_labels_list=labels, sampling=False)
    assert len(ref) == 7 and len(ref[0]) == 2 and ref[0][0].shape == (1344,)
    for b in range(2):
        pos = (ref[0][b] > 0).nonzero().view(-1)
        assert torch.equal(ref[5][b], pos)
        if b == 0 and pos.numel():
            assert torch.equal(ref[0][0][pos], labels[0][ref[6][0] - 1])
    assert ref[5][1].numel() == 0


@pytest.mark.parametrize("relu,C,sizes", [(True, 64, [(40, 40), (16, 16), (7, 9), (2, 2)]), (False, 64, [(40, 40), (16, 16), (7, 9), (2, 2)]),
                                          (True, 256, [(24, 24), (12, 12), (6, 6), (3, 3)]),      # groups spanning two chunks
                                          (True, 256, [(70, 70), (5, 5), (5, 5), (1, 1)])])
def test_group_norm_act_train_forward_backward_vs_torch(dev, relu, C, sizes):
    """The autograd GroupNorm(+ReLU) over a list of tensors (orp_groupnorm_act_multi_train / _backward): outputs, grad_input,
    dgamma and dbeta against torch.nn.GroupNorm (+ relu) per tensor; a module shared by several tensors gets the summed
    parameter gradients; two identical calls give identical bits."""
    import torch.nn as nn
    from orientedreppoints_amd.mmdet_ops.fused_norm import group_norm_act_train
    torch.manual_seed(0)
    B, G = 2, 32
    shared, own = nn.GroupNorm(G, C).to(dev), nn.GroupNorm(G, C).to(dev)
    with torch.no_grad():
        for m in (shared, own):
            m.weight.normal_(1.0, 0.3); m.bias.normal_(0, 0.3)
    mods = [shared, shared, own, shared]
    xs = [(torch.randn(B, C, h, w, device=dev) * 2 + 0.5).requires_grad_(True) for h, w in sizes]
    wts = [torch.randn(B, C, h, w, device=dev) for h, w in sizes]

    def run(fn):
        for t in xs + [shared.weight, shared.bias, own.weight, own.bias]:
            t.grad = None
        ys = fn()
        sum((y * w).sum() for y, w in zip(ys, wts)).backward()
        return [y.detach().clone() for y in ys], [x.grad.clone() for x in xs], \
            [p.grad.clone() for p in (shared.weight, shared.bias, own.weight, own.bias)]
    mine = run(lambda: group_norm_act_train(xs, mods, relu=relu))
    again = run(lambda: group_norm_act_train(xs, mods, relu=relu))
    ref = run(lambda: [torch.relu(m(x)) if relu else m(x) for m, x in zip(mods, xs)])
    for a, b in zip(mine[0] + mine[1] + mine[2], again[0] + again[1] + again[2]):
        assert torch.equal(a, b)
    for a, b in zip(mine[0], ref[0]):
        assert float((a - b).abs().max()) <= 2e-5
    for a, b in zip(mine[1], ref[1]):
        assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max()))
    for a, b in zip(mine[2], ref[2]):
        assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))


def _loss_rows(dev, P, seed):
    """P (point set, gt quad) rows: point sets scattered around quads of mixed size, some far outside, some inside."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from orientedreppoints_amd import synthetic as S
    rng = np.random.RandomState(seed)
    gt = S.gen_polys(P, seed, wh=(8, 60))[:, :8].astype(np.float32) / 8.0
    c = gt.reshape(P, 4, 2).mean(1)
    spread = rng.uniform(0.5, 12.0, size=(P, 1, 1)).astype(np.float32)
    pts = (c[:, None, :] + rng.normal(size=(P, 9, 2)).astype(np.float32) * spread).reshape(P, 18)
    return torch.from_numpy(pts).to(dev), torch.from_numpy(gt).to(dev)


@pytest.mark.parametrize("P,nseg", [(700, 5), (33, 1), (1, 3), (0, 5), (4100, 5)])
def test_segment_losses_equal_the_tensor_op_composition(dev, P, nseg):
    """train_ops._SegmentGIoULoss / segment_border_loss (rows kernel + one fixed-order segment reduction each) against the
    plain PyTorch composition the reference's GIoULoss / SpatialBorderLoss amount to per segment (tests/cpu_standins.py:
    the same convex_giou / pointsJf operators underneath, tensor operations around them): losses to 1e-6 of their scale,
    gradients to 1e-6, empty segments and zero weights included; twice the same bits."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import cpu_standins as CS
    from orientedreppoints_amd.mmdet_ops import train_ops
    from orientedreppoints_amd.mmdet_ops.iou_wrapper import convex_giou
    from orientedreppoints_amd.mmdet_ops.point_justify import points_in_quad_aligned
    pts, gt = _loss_rows(dev, max(P, 1), 3 + P)
    pts, gt = pts[:P], gt[:P]
    g = torch.Generator(device="cpu").manual_seed(P)
    seg = torch.randint(0, nseg, (P,), generator=g).to(dev)
    if nseg > 2:
        seg[seg == 1] = 0                                                     # an empty segment
    weight = (torch.rand(P, generator=g) > 0.25).float().to(dev)
    denom_int = torch.bincount(seg, minlength=nseg)                            # the init stage passes integer counts
    denom_f = torch.rand(nseg, generator=g).to(dev) * 40                       # the refine stage a float (may be < 1)
    for denom in (denom_int, denom_f):
        for lw in (0.375, 1.0):
            a = pts.clone().requires_grad_(True)
            b = pts.clone().requires_grad_(True)
            got = train_ops._SegmentGIoULoss.apply(a, gt, weight, seg, nseg, denom, lw)
            CS.SegmentGIoULossReference.giou_fn = staticmethod(convex_giou)
            want = CS.SegmentGIoULossReference.apply(b, gt, weight, seg, nseg, denom, lw)
            assert got.shape == want.shape == (nseg,)
            assert float((got - want).detach().abs().max()) <= 1e-6 * max(1.0, float(want.detach().abs().max()))
            if P:
                got.sum().backward(); want.sum().backward()
                assert float((a.grad - b.grad).abs().max()) <= 1e-6 * max(1.0, float(b.grad.abs().max()))
                # loss scaling (GradScaler.scale(loss).backward()): the incoming gradient must reach the stashed one --
                # a per-segment factor here; power-of-two scales reproduce the unscaled gradient's bits exactly
                c = pts.clone().requires_grad_(True)
                sc = torch.tensor([2.0 ** (k % 5) for k in range(nseg)], device=dev)
                (train_ops._SegmentGIoULoss.apply(c, gt, weight, seg, nseg, denom, lw) * sc).sum().backward()
                assert torch.equal(c.grad, a.grad * sc[seg].reshape((-1,) + (1,) * (a.grad.dim() - 1)))
            again = train_ops._SegmentGIoULoss.apply(pts, gt, weight, seg, nseg, denom, lw)
            assert torch.equal(again, got.detach())
            # border loss
            a = pts.clone().requires_grad_(True)
            b = pts.clone().requires_grad_(True)
            got = train_ops.segment_border_loss(a, gt, weight, seg, nseg, denom, lw)
            want = CS.segment_border_loss_reference(b, gt, weight, seg, nseg, denom, lw, points_in_quad_aligned)
            assert float((got - want).detach().abs().max()) <= 2e-6 * max(1.0, float(want.detach().abs().max()))
            coef = torch.arange(1, nseg + 1, device=dev, dtype=torch.float32)
            if P:
                (got * coef).sum().backward(); (want * coef).sum().backward()
                assert b.grad.abs().max() > 0 or weight.sum() == 0
                assert float((a.grad - b.grad).abs().max()) <= 2e-6 * max(1.0, float(b.grad.abs().max()))
            assert torch.equal(train_ops.segment_border_loss(pts, gt, weight, seg, nseg, denom, lw), got.detach())
