"""GPU regression of the round-2 driver failure (GPUTEST_r02: test_conv1x1_multi_vs_library_convolution, error 3.75):
a module is built, used, freed, and another of the SAME shape is built -- CPython hands out the old `id()`, the caching
allocator the old device address, `_version` and shape agree, and an `id()`-keyed cache served the dead module's pack.
Every op that caches a re-layout of a parameter (orientedreppoints_amd/_packcache.py) is driven through 50 such
generations with different weights; every generation's output is checked against an un-cached formulation."""
import gc

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import torch.nn as nn                    # noqa: E402
import torch.nn.functional as F          # noqa: E402

GENERATIONS = 50


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from orientedreppoints_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _recycled(counter, obj, seen):
    counter[0] += id(obj) in seen
    seen.add(id(obj))


def test_conv1x1_multi_generations(dev):
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv1x1_multi
    torch.manual_seed(0)
    xs = [torch.randn(2, 256, h, w, device=dev) for h, w in ((16, 16), (3, 5))]
    hits, seen = [0], set()
    for g in range(GENERATIONS):
        conv = nn.Conv2d(256, 18, 1).to(dev)
        _recycled(hits, conv.weight, seen)
        with torch.no_grad():
            ys = conv1x1_multi(xs, conv)
            for x, y in zip(xs, ys):
                want = F.conv2d(x, conv.weight, conv.bias)
                assert float((y - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), "generation %d" % g
        del conv, ys, y, want
    assert hits[0] > 0, "no id() was recycled: the regression was not exercised"


def test_deform_conv_and_conv3x3_generations(dev):
    import importlib
    dc = importlib.import_module("orientedreppoints_amd.mmdet_ops.deform_conv")
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv3x3_multi
    torch.manual_seed(1)
    x = torch.randn(1, 256, 12, 12, device=dev)
    off = torch.randn(1, 18, 12, 12, device=dev) * 1.5
    hits, seen = [0], set()
    for g in range(GENERATIONS):
        m = dc.DeformConv(256, 256, 3, padding=1).to(dev)
        c = nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
        _recycled(hits, m.weight, seen)
        with torch.no_grad():
            got = m(x, off)                                                    # packed MFMA path (cached pack)
            want = dc._forward_direct(x, off, None, m.weight, None, (1, 1), (1, 1), (1, 1), 1, 1)   # reads the weight itself
            assert float((got - want).abs().max()) <= 1e-4 * max(1.0, float(want.abs().max())), "DeformConv gen %d" % g
            got3 = conv3x3_multi([x], c)[0]
            want3 = F.conv2d(x, c.weight, None, 1, 1)
            assert float((got3 - want3).abs().max()) <= 1e-4 * max(1.0, float(want3.abs().max())), "conv3x3 gen %d" % g
        del m, c, got, want, got3, want3
    assert hits[0] > 0


@pytest.mark.parametrize("dtype", ["float16", "bfloat16"])
def test_half_deform_conv_generations(dev, dtype):
    import importlib
    dc = importlib.import_module("orientedreppoints_amd.mmdet_ops.deform_conv")
    dt = getattr(torch, dtype)
    torch.manual_seed(2)
    x = torch.randn(1, 256, 10, 10, device=dev).to(dt)
    off = (torch.randn(1, 18, 10, 10, device=dev) * 1.5).to(dt)
    tol = 4e-3 if dtype == "float16" else 3e-2
    for g in range(20):
        w = nn.Parameter((torch.randn(256, 256, 3, 3, device=dev) * 0.05).to(dt), requires_grad=False)
        got = dc.deform_conv_forward_multi_half([x], [off], w, 1, 1, 1)[0].float()
        want = dc._forward_direct(x.float(), off.float(), None, w.float(), None, (1, 1), (1, 1), (1, 1), 1, 1)
        assert float((got - want).abs().max()) <= tol * max(1.0, float(want.abs().max())), "half DCN gen %d" % g
        del w, got, want


def test_bn_affine_generations(dev):
    from orientedreppoints_amd.mmdet_ops.fused_norm import bn_act
    torch.manual_seed(3)
    x = torch.randn(2, 64, 9, 9, device=dev)
    hits, seen = [0], set()
    for g in range(GENERATIONS):
        bn = nn.BatchNorm2d(64).to(dev).eval()
        _recycled(hits, bn, seen)
        with torch.no_grad():
            bn.weight.normal_(1.0, 0.3); bn.bias.normal_(0, 0.3)
            bn.running_mean.normal_(0, 1.0); bn.running_var.uniform_(0.3, 2.0)
            assert float((bn_act(x.clone(), bn, relu=True) - torch.relu(bn(x))).abs().max()) <= 1e-5, "gen %d" % g
        del bn
    assert hits[0] > 0


def test_training_forward_sees_data_writes(dev):
    """With a trainable weight the pack is rebuilt on every call (cache_pack = not needs_input_grad): a write through
    `.data` -- no version bump -- is visible to the very next forward, so forward and backward use the same weight."""
    import importlib
    dc = importlib.import_module("orientedreppoints_amd.mmdet_ops.deform_conv")
    torch.manual_seed(4)
    m = dc.DeformConv(256, 256, 3, padding=1).to(dev)
    x = torch.randn(1, 256, 8, 8, device=dev, requires_grad=True)
    off = torch.randn(1, 18, 8, 8, device=dev)
    y0 = m(x, off)
    v0 = m.weight._version
    m.weight.data.mul_(2.0)
    assert m.weight._version == v0                         # the write was invisible to the version counter
    y1 = m(x, off)
    assert float((y1 - 2.0 * y0).detach().abs().max()) <= 1e-4 * max(1.0, float(y0.detach().abs().max()))
    # inference on a frozen weight caches; `.data` writes then need invalidate_packed_weights()
    m.weight.requires_grad_(False)
    with torch.no_grad():
        z0 = m(x, off)
        m.weight.data.mul_(0.5)
        dc.invalidate_packed_weights()
        z1 = m(x, off)
    assert float((z1 - 0.5 * z0).abs().max()) <= 1e-4 * max(1.0, float(z0.abs().max()))


def test_graphed_inference_after_model_rebuild(dev):
    """del model; model = build(...) with other weights: eager fused inference and a NEW GraphedInference must follow the
    new weights (packs and BN affines of the dead model must not be served)."""
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, GraphedInference, build_detector
    metas = [dict(img_shape=(128, 128, 3), pad_shape=(128, 128, 3), scale_factor=1.0, flip=False)]
    img = torch.randn(1, 3, 128, 128, device=dev, generator=torch.Generator(device=dev).manual_seed(7))

    def build(seed):
        torch.manual_seed(seed)
        model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
        head = model.bbox_head
        with torch.no_grad():
            head.reppoints_cls_out.weight.normal_(0, 0.05)
            head.reppoints_cls_out.bias.fill_(-3.3)
            # well-spread point sets (as test_graphed_inference_equals_simple_test): near-coincident points make the
            # min-area rectangle a tie-break between library-algorithm-level float differences
            head.reppoints_pts_init_out.bias.copy_(torch.tensor(
                [[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
                dtype=torch.float32, device=dev).reshape(-1) * 2.0)
            for mod in model.modules():
                if isinstance(mod, nn.BatchNorm2d):
                    mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.8, 1.2)
        return model

    for gen in range(3):
        model = build(100 + gen)
        gi = GraphedInference(model, img, metas)
        got = gi(img)
        with torch.no_grad():
            want = model.simple_test_batch(img, metas)
            # un-cached check of the head's logits: fused (packs / affines) vs grad-enabled stock modules
            fused = model.bbox_head(model.extract_feat(img))[0]
        with torch.enable_grad():
            ref = model.bbox_head(model.extract_feat(img))[0]
        for a, b in zip(fused, ref):
            assert float((a - b.detach()).abs().max()) <= 2e-3 * max(1.0, float(b.detach().abs().max())), "generation %d" % gen
        for gr, wr in zip(got, want):
            assert [c.shape for c in gr] == [c.shape for c in wr]
            for a, b in zip(gr, wr):
                assert np.allclose(a, b, rtol=1e-4, atol=1e-2)
        del model, gi, got, want, fused, ref
        gc.collect()
