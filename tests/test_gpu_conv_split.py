"""GPU: the head's tower convolutions on the bf16 matrix pipe (csrc/orp_conv_split.hip = the PLAIN instantiation of
csrc/orp_dcn_split.hip: every fp32 operand split exactly into three bf16 pieces, 6 or 9 partial products, fp32 accumulation)
and the channels-last GroupNorm(+ReLU) that sits between two of them (csrc/orp_norm.hip gn_cl_*).  Reference operator:
ConvModule.forward (mmdet/ops/conv_module.py:130-140: conv -> GroupNorm -> ReLU) as the head builds it
(mmdet/models/anchor_heads/orientedreppoints_head.py:91-113).  Checker: the oracle's DeformConv forward with ZERO offsets (the
reference's float samples are then the pixels themselves, deform_conv_cuda_kernel.cu:84-115, contracted in DOUBLE) and
torch's float64 convolution / GroupNorm on the CPU.  Tolerance 1e-5 of the output scale (north_star: 1e-4)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _needs_split_mode():
    """The automatic routes these tests assert are off by configuration when the suite runs under ORP_DCN_SPLIT=0 (the exact-fp32
    DeformConv kernel, library convolutions for towers / FPN); the explicit-mode tests of this file do not depend on the default."""
    from orientedreppoints_amd import _lib
    if _lib.lib().orp_dcn_get_split_mode() == 0:
        pytest.skip("ORP_DCN_SPLIT=0: the split convolution routes are switched off")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from orientedreppoints_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _cl(t):
    return t.contiguous(memory_format=torch.channels_last)


def _conv(cin, cout, k, dev, stride=1, pad=None, dil=1, bias=False, seed=0, std=0.05):
    g = torch.Generator().manual_seed(seed)
    m = nn.Conv2d(cin, cout, k, stride=stride, padding=(dil * (k - 1) // 2 if pad is None else pad), dilation=dil, bias=bias)
    with torch.no_grad():
        m.weight.copy_(torch.randn(m.weight.shape, generator=g) * std)
        if bias:
            m.bias.copy_(torch.randn(cout, generator=g))
    return m.to(dev).eval()


def _rel(got, want):
    want = want.double()
    return float((got.double().cpu() - want.cpu()).abs().max() / max(1e-6, float(want.abs().max())))


@pytest.mark.parametrize("B,C,H,W,Cout", [(2, 256, 16, 16, 256), (1, 64, 9, 11, 64), (1, 128, 5, 40, 192), (3, 256, 7, 9, 256)])
def test_conv_split_vs_oracle_and_float64(dev, oracle, B, C, H, W, Cout):
    import conftest
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi
    torch.manual_seed(C + H)
    conv = _conv(C, Cout, 3, dev, seed=C + W)
    x = torch.randn(B, C, H, W)
    want64 = F.conv2d(x.double(), conv.weight.detach().cpu().double(), padding=1)
    want_orc = oracle.dcn_forward(x.numpy(), np.zeros((B, 18, H, W), np.float32), conv.weight.detach().cpu().numpy(),
                                  stride=1, pad=1, dil=1)
    assert float(np.abs(want_orc - want64.numpy()).max()) <= 1e-6 * float(want64.abs().max())    # the two checkers agree
    errs = {}
    with torch.no_grad():
        lib = F.conv2d(x.to(dev), conv.weight, padding=1)
        errs['library'] = _rel(lib, want64)
        for nprod in (9, 6, 3):
            got = conv_split_multi([_cl(x.to(dev))], conv, nprod=nprod)[0]
            assert got.shape == want64.shape and got.is_contiguous(memory_format=torch.channels_last)
            errs[nprod] = _rel(got, want64)
            assert errs[nprod] <= 1e-5, nprod
            nchw = conv_split_multi([_cl(x.to(dev))], conv, nprod=nprod, out_channels_last=False)[0]
            assert nchw.is_contiguous() and torch.equal(nchw, got.contiguous()), "NCHW / NHWC outputs: same bits"
            assert torch.equal(conv_split_multi([_cl(x.to(dev))], conv, nprod=nprod)[0], got)
    conftest.REPORT.append("3x3 convolution %dx%dx%dx%d -> %d, max |err| / max |out| vs float64: library fp32 %.2e, split 9 products "
                           "%.2e, split 6 products %.2e, two fp16 pieces / 3 products %.2e"
                           % (B, C, H, W, Cout, errs['library'], errs[9], errs[6], errs[3]))
    # what remains is the fp32 accumulator's rounding (one per 16-channel MFMA and product, in a chain over K = 9 Cin): the
    # same figure the DeformConv forward has on both of its paths (test_gpu_dcn_split.py); six products lose nothing to nine
    assert errs[6] <= errs[9] + 5e-8 and errs[3] <= 2.0 * max(errs[9], errs['library']) + 1e-7


def test_conv_split_pair_levels_bias_relu_strides(dev):
    """Two layers in one launch == two single launches bit for bit; several levels per launch; bias + ReLU epilogue; stride 2,
    dilation 2, 1 x 1 and 1 x 3 kernels, asymmetric padding -- against float64."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi
    torch.manual_seed(11)
    shapes = [(24, 20), (12, 10), (6, 5), (3, 3), (1, 2)]
    for B in (1, 2):
        xa = [torch.randn(B, 256, h, w, device=dev) for h, w in shapes]
        xb = [torch.randn(B, 256, h, w, device=dev) for h, w in shapes]
        ca, cb = _conv(256, 256, 3, dev, bias=True, seed=1), _conv(256, 256, 3, dev, bias=True, seed=2)
        with torch.no_grad():
            pa, pb = conv_split_multi([_cl(t) for t in xa], ca, [_cl(t) for t in xb], cb, bias=True, relu=True)
            sa = conv_split_multi([_cl(t) for t in xa], ca, bias=True, relu=True)
            sb = conv_split_multi([_cl(t) for t in xb], cb, bias=True, relu=True)
            for u, v in zip(pa + pb, sa + sb):
                assert torch.equal(u, v)
            for u, x, c in zip(pa + pb, xa + xb, [ca] * len(xa) + [cb] * len(xb)):
                want = F.relu(F.conv2d(x.double().cpu(), c.weight.double().cpu(), c.bias.double().cpu(), padding=1))
                assert _rel(u, want) <= 1e-5
            # same input for both layers (the towers' first layer reads the FPN output twice)
            qa, qb = conv_split_multi([_cl(t) for t in xa], ca, [_cl(t) for t in xa], cb)
            for u, x in zip(qb, xa):
                assert _rel(u, F.conv2d(x.double().cpu(), cb.weight.double().cpu(), padding=1)) <= 1e-5
    x = torch.randn(2, 128, 13, 17, device=dev)
    for k, stride, pad, dil in ((3, 2, 1, 1), (3, 1, 2, 2), (1, 1, 0, 1), (3, 2, 0, 1), ((1, 3), 1, (0, 1), 1), (3, (2, 1), (1, 0), 1)):
        m = nn.Conv2d(128, 64, k, stride=stride, padding=pad, dilation=dil, bias=True).to(dev).eval()
        with torch.no_grad():
            for fmt in (True, False):
                got = conv_split_multi([_cl(x)], m, bias=True, out_channels_last=fmt)[0]
                want = F.conv2d(x.double().cpu(), m.weight.double().cpu(), m.bias.double().cpu(), stride=stride, padding=pad, dilation=dil)
                assert got.shape == want.shape
                assert _rel(got, want) <= 1e-5, (k, stride, pad, dil)


def test_conv_split_is_exact_on_integers_and_rejects_bad_arguments(dev):
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi, conv_split_ok
    g = torch.Generator().manual_seed(1)
    x = torch.randint(-8, 9, (1, 64, 12, 12), generator=g).float().to(dev)
    m = nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev)
    with torch.no_grad():
        m.weight.copy_(torch.randint(-4, 5, (64, 64, 3, 3), generator=g).float())
        want = F.conv2d(x.double(), m.weight.double(), padding=1)
        for nprod in (9, 6, 3):
            assert torch.equal(conv_split_multi([_cl(x)], m, nprod=nprod)[0].double(), want)
        assert conv_split_ok(m, x)
        assert not conv_split_ok(nn.Conv2d(48, 64, 3, padding=1).to(dev))
        assert not conv_split_ok(nn.Conv2d(64, 64, 5, padding=2).to(dev))
        with pytest.raises(ValueError):
            conv_split_multi([x], m)                                   # NCHW memory
        with pytest.raises(_lib.OrpHipError):
            conv_split_multi([_cl(x)], m, nprod=7)


@pytest.mark.parametrize("C,G", [(256, 32), (64, 8), (128, 32), (1024, 32)])
def test_groupnorm_channels_last_vs_float64(dev, C, G):
    from orientedreppoints_amd.mmdet_ops.fused_norm import group_norm_act_multi, group_norm_act_multi_cl
    torch.manual_seed(C)
    shapes = [(20, 24), (7, 9), (3, 3), (1, 2), (33, 31)]
    for B in (1, 3):
        gns = []
        for i in range(len(shapes)):
            gn = nn.GroupNorm(G, C).to(dev)
            with torch.no_grad():
                gn.weight.copy_(torch.randn(C) * 0.5 + 1.0); gn.bias.copy_(torch.randn(C) * 0.3)
            gns.append(gn)
        xs = [torch.randn(B, C, h, w, device=dev) * (1.0 + i) + 3.0 * i for i, (h, w) in enumerate(shapes)]
        with torch.no_grad():
            for relu in (True, False):
                got = group_norm_act_multi_cl([_cl(x) for x in xs], gns, relu=relu, inplace=False)
                for x, gn, y in zip(xs, gns, got):
                    assert y.is_contiguous(memory_format=torch.channels_last)
                    want = F.group_norm(x.double().cpu(), G, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps)
                    want = F.relu(want) if relu else want
                    assert float((y.double().cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
            # one module for all tensors, in place; close to the NCHW kernel pair (different summation grouping)
            ins = [_cl(x) for x in xs]
            outs = group_norm_act_multi_cl(ins, gns[0], relu=True)
            ref = group_norm_act_multi([x.clone() for x in xs], gns[0], relu=True, inplace=False)
            for a, b, c in zip(outs, ins, ref):
                assert a.data_ptr() == b.data_ptr()
                assert float((a - c).abs().max()) <= 1e-5 * max(1.0, float(c.abs().max()))
            again = group_norm_act_multi_cl([_cl(x) for x in xs], gns[0], relu=True)
            for a, b in zip(outs, again):
                assert torch.equal(a, b)


def test_to_channels_last_multi(dev):
    from orientedreppoints_amd.mmdet_ops.fused_norm import to_channels_last_multi
    torch.manual_seed(2)
    xs = [torch.randn(2, 96, h, w, device=dev) for h, w in ((17, 33), (5, 5), (1, 1), (64, 3))]
    xs[1] = _cl(xs[1])
    outs = to_channels_last_multi(xs)
    assert outs[1] is xs[1]
    for x, y in zip(xs, outs):
        assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last) and torch.equal(x, y)


def test_head_inference_with_channels_last_towers_vs_reference_forward(dev):
    """OrientedRepPointsHead.forward at inference with both towers on the channels-last path (to_channels_last ->
    [pair convolution -> GroupNorm+ReLU] x 3 -> init convolution with bias + ReLU -> DeformConv pair gathering the towers'
    outputs as they are) against the reference's per-level forward_single (head :148-171: stock modules, one level at a time)
    and against the library-convolution towers; B = 1 and 2; off when the library's split mode is off."""
    import conftest
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.dota_configs import r50_model
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.registry import build_head
    _needs_split_mode()
    torch.manual_seed(9)
    head = build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()
    with torch.no_grad():
        head.reppoints_pts_init_out.weight.normal_(0, 0.05)
        for m in list(head.cls_convs) + list(head.reg_convs):
            m.conv.weight.normal_(0, 0.03); m.norm.weight.uniform_(0.5, 1.5); m.norm.bias.normal_(0, 0.2)
        for B in (1, 2):
            feats = [torch.randn(B, 256, h, w, device=dev) for h, w in ((40, 36), (20, 18), (10, 9), (5, 5), (3, 2))]
            head.split_towers = None
            assert head._split_towers_ok(feats)                 # automatic mode: on at every pyramid size, down to 3 x 2 levels
            got = head(feats)
            head.split_towers = False
            lib = head(feats)
            want = [head.forward_single(f) for f in feats]
            worst = 0.0
            for k in range(3):                                   # cls_out, pts_out_init, pts_out_refine
                for lvl in range(len(feats)):
                    g, l, w = got[k][lvl], lib[k][lvl], want[lvl][k]
                    assert g.shape == w.shape and g.is_contiguous()
                    scale = max(1.0, float(w.abs().max()))
                    worst = max(worst, float((g - w).abs().max()) / scale)
                    assert float((g - w).abs().max()) <= 1e-4 * scale, (k, lvl)
                    assert float((g - l).abs().max()) <= 1e-4 * scale, (k, lvl)
            conftest.REPORT.append("head inference, channels-last towers vs forward_single, B = %d: max |diff| / scale %.2e" % (B, worst))
        big = [torch.randn(1, 256, n, n, device=dev) for n in (64, 32, 16, 8, 8)]
        head.split_towers = None
        assert head._split_towers_ok(big)
        L = _lib.lib()
        L.orp_dcn_set_split_mode(0)
        try:
            head.split_towers = True
            assert not head._split_towers_ok(big)                # exact-fp32 mode: the library-convolution towers
        finally:
            L.orp_dcn_set_split_mode(-1)
            head.split_towers = None


def test_conv_split_at_a_head_level_vs_float64_and_the_library(dev):
    """One 256 -> 256 layer at a 64^2 level (the library's pick there is Winograd F(2x2, 3x3)): both against torch's float64
    convolution on the CPU; the split path's error is reported next to the library's and must be of the same order."""
    import conftest
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi
    torch.manual_seed(21)
    conv = _conv(256, 256, 3, dev, seed=5, std=0.03)
    x = torch.randn(1, 256, 64, 64) * 2.0
    want = F.conv2d(x.double(), conv.weight.detach().cpu().double(), padding=1)
    with torch.no_grad():
        e_lib = _rel(F.conv2d(x.to(dev), conv.weight, padding=1), want)
        e6 = _rel(conv_split_multi([_cl(x.to(dev))], conv, nprod=6)[0], want)
        e9 = _rel(conv_split_multi([_cl(x.to(dev))], conv, nprod=9)[0], want)
        e3 = _rel(conv_split_multi([_cl(x.to(dev))], conv, nprod=3)[0], want)
        # inputs spanning 2^40: the range scaling must hold (no overflow to inf, small values keep their absolute precision)
        xw = x * torch.exp2(torch.randint(-30, 11, x.shape).float())
        want_w = F.conv2d(xw.double(), conv.weight.detach().cpu().double(), padding=1)
        e3w = _rel(conv_split_multi([_cl(xw.to(dev))], conv, nprod=3)[0], want_w)
        e6w = _rel(conv_split_multi([_cl(xw.to(dev))], conv, nprod=6)[0], want_w)
    conftest.REPORT.append("3x3 convolution 1x256x64x64 -> 256, max |err| / max |out| vs float64: library fp32 %.2e, split 9 products "
                           "%.2e, split 6 products %.2e, two fp16 pieces %.2e; inputs spanning 2^40: 6 products %.2e, fp16 pieces %.2e"
                           % (e_lib, e9, e6, e3, e6w, e3w))
    assert e6 <= 1e-5 and e9 <= 1e-5 and e3 <= 1e-5 and e3w <= 1e-5
    if os.environ.get("ORP_DCNS_WS", "0") == "1":
        # the wave-specialised kernel (a dev switch, csrc/orp_dcn_split.hip) has no side accumulators: 1.28e-6 here, which is WHY it is
        # not the default; under the switch the suite reports the figure instead of failing on the default kernel's gate
        conftest.REPORT.append("ORP_DCNS_WS=1: fp16-pieces convolution error %.3e against the default kernel's gate %.3e" % (e3, 1.5 * e_lib + 5e-8))
        return
    assert e3 <= 1.5 * e_lib + 5e-8
    # the same order as the library's own fp32 convolution at this shape (measured 8.8e-7 against 7.7e-7; 2.0e-6 before the
    # small partial products got their own accumulator set)
    assert e6 <= 1.5 * e_lib + 5e-8


def test_fpn_output_convolutions_one_layer_per_level(dev):
    """FPN.forward at inference (mmdet/models/necks/fpn.py:118-175): the output convolutions of all levels -- a different
    ConvModule per level -- as one orp_conv_split_multi_ex launch, channels-last results; against the module-by-module path
    (stock ConvModules) and the library-convolution path; B = 1 and 2, odd sizes."""
    from orientedreppoints_amd.dota_configs import r50_model
    from orientedreppoints_amd.mmdet_models import ConfigDict
    from orientedreppoints_amd.mmdet_models.registry import build_neck
    torch.manual_seed(13)
    neck = build_neck(ConfigDict(r50_model['neck'])).to(dev).eval()
    with torch.no_grad():
        for m in list(neck.lateral_convs) + list(neck.fpn_convs):
            m.conv.weight.normal_(0, 0.03)
            if getattr(m, 'norm', None) is not None:
                m.norm.weight.uniform_(0.5, 1.5); m.norm.bias.normal_(0, 0.2)
        for B, sizes in ((1, (64, 32, 16, 8)), (2, (44, 22, 11, 6))):
            inputs = [torch.randn(B, c, n, n, device=dev) for c, n in zip(neck.in_channels, sizes)]
            neck.split_convs = True
            got = neck(inputs)
            neck.split_convs = False
            lib = neck(inputs)
            # module by module: what ConvModule.forward does (conv -> norm), the reference's statement order
            used = len(neck.lateral_convs)
            lats = [lc(inputs[i + neck.start_level]) for i, lc in enumerate(neck.lateral_convs)]
            for i in range(used - 1, 0, -1):
                lats[i - 1] = lats[i - 1] + F.interpolate(lats[i], size=lats[i - 1].shape[2:], mode='nearest')
            want = [neck.fpn_convs[i](lats[i]) for i in range(used)]
            assert len(got) == len(lib) == neck.num_outs
            for i in range(used):
                assert got[i].shape == want[i].shape
                scale = max(1.0, float(want[i].abs().max()))
                assert float((got[i] - want[i]).abs().max()) <= 1e-4 * scale
                assert float((got[i] - lib[i]).abs().max()) <= 1e-4 * scale
            for i in range(used, neck.num_outs):                 # the extra levels do not depend on the output convolutions here
                assert float((got[i] - lib[i]).abs().max()) <= 1e-4 * max(1.0, float(lib[i].abs().max()))


def test_conv_split_train_gradients_vs_float64(dev):
    """conv_split_train: one autograd node for the levels of a tower layer (one weight), the two towers' layer k (two
    weights, first / second half of the tensors, the first layer reading the SAME tensors twice) and the FPN's output
    convolutions (a weight per tensor): outputs, grad_input and grad_weight against torch's float64 convolution on the CPU
    (grad_input = the same kernel with the flipped, transposed weights; grad_weight = the library's kernel per level)."""
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_train, conv_split_train_ok
    from orientedreppoints_amd import switches
    _needs_split_mode()
    if not switches.TRAIN_SPLIT:
        pytest.skip("ORP_TRAIN_SPLIT=0: the training convolutions are routed to the library")
    torch.manual_seed(17)
    shapes = [(12, 10), (6, 5), (3, 3)]
    B, C = 2, 128
    convs = [nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev) for _ in range(3)]
    with torch.no_grad():
        for c in convs:
            c.weight.normal_(0, 0.04)
    assert conv_split_train_ok(convs, torch.zeros(1, C, 4, 4, device=dev))

    def reference(xs, mods, gos):
        ws = {id(m): m.weight.detach().double().cpu().requires_grad_(True) for m in mods}
        xr = {}
        outs = []
        for x, m in zip(xs, mods):
            xd = xr.setdefault(id(x), x.detach().double().cpu().requires_grad_(True))
            outs.append(F.conv2d(xd, ws[id(m)], padding=1))
        loss = sum((o * g.double().cpu()).sum() for o, g in zip(outs, gos))
        loss.backward()
        return outs, xr, ws

    def check(xs, mods):
        xs_g = {}
        ins = []
        for x in xs:
            ins.append(xs_g.setdefault(id(x), x.clone().requires_grad_(True)))
        for m in mods:
            m.weight.grad = None
        outs = conv_split_train(ins, mods)
        gos = [torch.randn_like(o) for o in outs]
        sum((o * g).sum() for o, g in zip(outs, gos)).backward()
        r_outs, r_x, r_w = reference(xs, mods, gos)
        for o, r in zip(outs, r_outs):
            assert o.is_contiguous() and _rel(o, r.detach()) <= 1e-5
        for x in {id(x): x for x in xs}.values():
            assert _rel(xs_g[id(x)].grad, r_x[id(x)].grad) <= 1e-5
        for m in {id(m): m for m in mods}.values():
            assert _rel(m.weight.grad, r_w[id(m)].grad) <= 1e-4

    xs = [torch.randn(B, C, h, w, device=dev) for h, w in shapes]
    check(xs, [convs[0]] * 3)                                        # one layer, all levels
    check(xs + xs, [convs[0]] * 3 + [convs[1]] * 3)                  # two layers reading the same tensors (towers' layer 0)
    ys = [torch.randn(B, C, h, w, device=dev) for h, w in shapes]
    check(xs + ys, [convs[0]] * 3 + [convs[1]] * 3)                  # two layers, own inputs
    check(xs, convs)                                                 # a layer per tensor (FPN)
    # no gradient wanted for the inputs: still the weights'
    for c in convs:
        c.weight.grad = None
    outs = conv_split_train([x.detach() for x in xs], convs[0])
    sum(o.sum() for o in outs).backward()
    assert convs[0].weight.grad is not None


def test_range_hand_over_between_producers_and_the_fp16_pieces_convolution(dev):
    """The fp16-pieces mode scales its samples by a power of two taken from max |x| of the inputs.  Producers can leave that on
    the device (`Amax`): the transposition the exact maximum per slot, the channels-last GroupNorm an upper bound from its
    statistics pass -- never below the true maximum, and not wastefully above it; a convolution run with such a bound gives the
    float64 result to the same tolerance as with its own pre-pass."""
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.mmdet_ops.fused_norm import (Amax, conv_split_multi, group_norm_act_multi_cl,
                                                          to_channels_last_multi)
    L = _lib.lib()
    assert L.orp_dcn_set_split_mode(3) == 0              # producers leave ranges in the fp16-pieces mode only
    try:
        _range_hand_over_body(dev, Amax, conv_split_multi, group_norm_act_multi_cl, to_channels_last_multi)
    finally:
        L.orp_dcn_set_split_mode(-1)


def _range_hand_over_body(dev, Amax, conv_split_multi, group_norm_act_multi_cl, to_channels_last_multi):
    torch.manual_seed(23)
    xs = [torch.randn(2, 256, h, w, device=dev) * (1.0 + 3.0 * i) for i, (h, w) in enumerate(((20, 24), (7, 9), (3, 3), (16, 16)))]
    cl, bits = to_channels_last_multi(xs, amax_slots=[0, 0, 1, 1])
    got = bits.view(torch.float32).cpu()
    assert float(got[0]) == max(float(xs[0].abs().max()), float(xs[1].abs().max()))
    assert float(got[1]) == max(float(xs[2].abs().max()), float(xs[3].abs().max()))
    more = [torch.randn(2, 256, 5, 5, device=dev) * 100.0]
    _, bits2 = to_channels_last_multi(more, amax_into=(bits, [0]))                 # merged into slot 0, nothing reset
    assert bits2 is bits and float(bits.view(torch.float32)[0]) == float(more[0].abs().max())
    gn = nn.GroupNorm(32, 256).to(dev)
    with torch.no_grad():
        gn.weight.uniform_(0.2, 2.0); gn.bias.normal_(0, 0.5)
        for relu in (True, False):
            ys, b = group_norm_act_multi_cl([t.clone() for t in cl], gn, relu=relu, amax_slots=[0, 1, 1, 0])
            bound = b.view(torch.float32).cpu()
            for slot, idx in ((0, (0, 3)), (1, (1, 2))):
                true = max(float(ys[i].abs().max()) for i in idx)
                assert true <= float(bound[slot]) <= 8.0 * true, (relu, slot, true, float(bound[slot]))
        conv = _conv(256, 256, 3, dev, seed=3, std=0.03)
        ys, b = group_norm_act_multi_cl([t.clone() for t in cl], gn, relu=True, amax_slots=[0] * 4)
        with_bound = conv_split_multi(ys, conv, nprod=3, amax=Amax(b, 0))
        own = conv_split_multi(ys, conv, nprod=3)
        for y, u, v in zip(ys, with_bound, own):
            want = F.conv2d(y.double().cpu(), conv.weight.double().cpu(), padding=1)
            assert _rel(u, want) <= 2e-6 and _rel(v, want) <= 2e-6


def test_conv_weight_gradient_kernel_vs_float64(dev):
    """orp_conv_wgrad_split: grad_weight of a 256 -> 256 stride-1 'same' convolution over several levels in one launch, NCHW
    tensors, fp16-pieces arithmetic -- against torch's float64 weight gradient on the CPU; 3x3, dilated 3x3 and 1x1 kernels,
    odd sizes, B = 1 and 2, with the ranges taken by the kernel's own pre-pass and handed in; bitwise reproducible; and
    through conv_split_train (the autograd node) for 256-channel layers."""
    import conftest
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_train, conv_wgrad_split
    torch.manual_seed(29)
    # (W = 24 / 8: octets that hang over a row end by one position; 9, 3, 2: wrapping octets; the second set -- every W a multiple of
    #  8, every H * W of 32 -- takes the branch-free, deeper-pipelined instantiation for the 3 x 3 and 1 x 1 kernels)
    shape_sets = ([(20, 24), (7, 9), (3, 3), (1, 2), (16, 8)], [(16, 8), (8, 16), (4, 8), (32, 32)])
    worst = 0.0
    for B, shapes in ((1, shape_sets[0]), (2, shape_sets[0]), (1, shape_sets[1]), (2, shape_sets[1])):
        for k, pad, dil in ((3, 1, 1), (3, 2, 2), (1, 0, 1)):
            xs = [torch.randn(B, 256, h, w, device=dev) * (1.0 + i) for i, (h, w) in enumerate(shapes)]
            gs = [torch.randn(B, 256, h, w, device=dev) * 0.01 * (1.0 + i) for i, (h, w) in enumerate(shapes)]
            want = sum(torch.nn.grad.conv2d_weight(x.double().cpu(), (256, 256, k, k), g.double().cpu(), padding=pad, dilation=dil)
                       for x, g in zip(xs, gs))
            got = conv_wgrad_split(xs, gs, (256, 256, k, k), (pad, pad), (dil, dil))
            assert got.shape == want.shape
            worst = max(worst, _rel(got, want))
            assert _rel(got, want) <= 1e-5, (B, k, pad, dil)
            assert torch.equal(conv_wgrad_split(xs, gs, (256, 256, k, k), (pad, pad), (dil, dil)), got)
            ax = torch.tensor([max(float(x.abs().max()) for x in xs) * 3.0], device=dev).view(torch.int32)   # a loose bound
            ag = torch.tensor([max(float(g.abs().max()) for g in gs)], device=dev).view(torch.int32)
            assert _rel(conv_wgrad_split(xs, gs, (256, 256, k, k), (pad, pad), (dil, dil), ax, ag), want) <= 1e-5
    conftest.REPORT.append("convolution weight gradient (256 -> 256, 4 - 5 levels, both instantiations), max |err| / max |grad| vs float64: %.2e" % worst)
    # through the autograd node: two 256-channel layers on their own inputs (pair layout)
    convs = [nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev) for _ in range(2)]
    shapes = shape_sets[0]
    xa = [torch.randn(2, 256, h, w, device=dev).requires_grad_(True) for h, w in shapes[:3]]
    xb = [torch.randn(2, 256, h, w, device=dev).requires_grad_(True) for h, w in shapes[:3]]
    outs = conv_split_train(xa + xb, [convs[0]] * 3 + [convs[1]] * 3)
    gos = [torch.randn_like(o) for o in outs]
    sum((o * g).sum() for o, g in zip(outs, gos)).backward()
    for c, xs_, gg in ((convs[0], xa, gos[:3]), (convs[1], xb, gos[3:])):
        want = sum(torch.nn.grad.conv2d_weight(x.detach().double().cpu(), (256, 256, 3, 3), g.double().cpu(), padding=1)
                   for x, g in zip(xs_, gg))
        assert _rel(c.weight.grad, want) <= 1e-5


@pytest.mark.parametrize("mode", [6, 3])
def test_tower_layers_with_groupnorm_fused_around_the_convolutions(dev, mode):
    """conv -> GroupNorm -> ReLU -> conv -> GroupNorm -> ReLU of both towers over a five-level pyramid with the normalisation fused
    around the convolution launches (`conv_split_gn`: statistics from the convolution's epilogue, affine + ReLU applied by the next
    layer as it reads, the last one materialised) against the same modules in float64 on the CPU (reference head :91-113,
    ConvModule: conv, norm, activate) and against the unfused path (`conv_split_multi` + `group_norm_act_multi_cl`); B = 1, 2 --
    tiles never span two images --, levels down to 1 x 3; twice: the results are bitwise reproducible."""
    import conftest
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.mmdet_ops.fused_norm import (Amax, conv_split_gn, conv_split_gn_ok, conv_split_multi, group_norm_act_multi_cl,
                                                            to_channels_last_multi)
    L = _lib.lib()
    torch.manual_seed(31)
    convs = [[_conv(256, 256, 3, dev, seed=40 + 2 * k + t, std=0.03) for t in range(2)] for k in range(2)]
    gns = [[nn.GroupNorm(32, 256).to(dev) for t in range(2)] for k in range(2)]
    with torch.no_grad():
        for row in gns:
            for g in row:
                g.weight.uniform_(0.5, 1.5); g.bias.normal_(0, 0.2)
    assert L.orp_dcn_set_split_mode(mode) == 0
    worst = 0.0
    try:
        with torch.no_grad():
            for B in (1, 2):
                xs = [torch.randn(B, 256, h, w, device=dev) * (1.0 + 0.5 * i) for i, (h, w) in enumerate(((24, 20), (12, 10), (6, 5), (3, 3), (1, 3)))]
                n = len(xs)
                assert conv_split_gn_ok(convs[0][0], convs[0][1], gns[0][0], gns[0][1], xs[0])
                cl, bits = to_channels_last_multi(xs, amax_slots=[0] * n)
                am0 = Amax(bits, 0) if bits is not None else None

                def fused():
                    a, b, coef, am = conv_split_gn(cl, convs[0][0], cl, convs[0][1], gns[0][0], gns[0][1], amax=am0)
                    assert coef is not None and tuple(coef.shape) == (2 * n, B, 256, 2)
                    a, b, coef, am = conv_split_gn(a, convs[1][0], b, convs[1][1], gns[1][0], gns[1][1], coef_in=coef, amax=am, materialize=True)
                    assert coef is None and (am is not None) == (mode == 3)
                    return a + b
                got = fused()
                again = fused()
                for x, y in zip(got, again):
                    assert torch.equal(x, y)
                # the unfused path
                oa, ob = conv_split_multi(cl, convs[0][0], cl, convs[0][1], amax=am0)
                both, bits1 = group_norm_act_multi_cl(oa + ob, [gns[0][0]] * n + [gns[0][1]] * n, relu=True, amax_slots=[0] * n + [1] * n)
                oa, ob = conv_split_multi(both[:n], convs[1][0], both[n:], convs[1][1], amax=Amax(bits1, 1) if bits1 is not None else None)
                unf, _ = group_norm_act_multi_cl(oa + ob, [gns[1][0]] * n + [gns[1][1]] * n, relu=True, amax_slots=[0] * n + [1] * n)
                for t in range(2):
                    for i in range(n):
                        x64 = xs[i].double().cpu()
                        for k in range(2):
                            c, g = convs[k][t], gns[k][t]
                            x64 = F.conv2d(x64, c.weight.detach().double().cpu(), padding=1)
                            x64 = F.relu(F.group_norm(x64, 32, g.weight.detach().double().cpu(), g.bias.detach().double().cpu(), g.eps))
                        gt = got[t * n + i]
                        assert gt.shape == x64.shape and gt.is_contiguous(memory_format=torch.channels_last)
                        scale = float(x64.abs().max())
                        e = float((gt.double().cpu() - x64).abs().max()) / scale
                        worst = max(worst, e)
                        assert e <= 2e-5, (B, t, i, e)
                        assert float((gt - unf[t * n + i]).abs().max()) <= 2e-5 * scale
    finally:
        L.orp_dcn_set_split_mode(-1)
    conftest.REPORT.append("two tower layers with fused GroupNorm (mode %d), max |err| / max |out| vs float64 modules: %.2e" % (mode, worst))


def _small_detector(dev):
    from orientedreppoints_amd.dota_configs import r50_model, test_cfg
    from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
    torch.manual_seed(0)
    model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
    with torch.no_grad():
        model.bbox_head.reppoints_cls_out.weight.normal_(0, 0.05)
        model.bbox_head.reppoints_cls_out.bias.fill_(-3.3)
    return model


def test_fp16_pieces_mode_graph_replay_after_an_eager_call(dev):
    """Round 4's open issue, now a regression test: fp16-pieces arithmetic, small pyramid (the automatic mode takes the split
    path at every size), replay / eager call / replay.  The replays used to return no detections: the range words were zeroed
    by hipMemsetAsync NODES, which wrote 0x80808080 in replays that followed eager work (tests/checks/graph_bitwise.py with
    ORP_FILL=memset); every fill of the library is a kernel node now (csrc/orp_launch.hpp fill_async)."""
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.mmdet_models import GraphedInference
    L = _lib.lib()
    model = _small_detector(dev)
    metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)]
    assert L.orp_dcn_set_split_mode(3) == 0
    try:
        gi = GraphedInference(model, torch.randn(1, 3, 256, 256, device=dev), metas)
        for seed in (1, 2, 3, 4):
            img = torch.randn(1, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
            got = gi(img)
            with torch.no_grad():
                want = model.simple_test_batch(img, metas)                 # the eager call in between
            assert sum(len(c) for r in got for c in r) == sum(len(c) for r in want for c in r) > 0
            for gr, wr in zip(got, want):
                for a, b in zip(gr, wr):
                    assert a.shape == b.shape and np.array_equal(a, b)
    finally:
        L.orp_dcn_set_split_mode(-1)


@pytest.mark.parametrize("mode", [6, 3])
def test_concurrent_graph_replays_are_bitwise_the_eager_step_at_a_small_pyramid(dev, mode):
    """256^2 images, two per call (pyramid down to 2 x 2: tile height 1, the workgroups that share a CU), three captured graphs
    replayed CONCURRENTLY on their own streams, every replay's head outputs and detections compared BIT FOR BIT with the eager
    step of its images.  What round 4 could not pass: replays next to other streams' kernels differed in a few positions of the
    DeformConv output (all channels) and, rarely, in the rotated NMS's keep count -- packed-fp32 VALU instructions returning a
    wrong low half in lanes 48..63 while other waves ran dense MFMAs (tests/checks/mfma_refill_victim.hip);
    the library is built without those instructions (build.py NO_PACKED_FP32; tests/test_capi_symbols.py checks the binary)."""
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.mmdet_models import GraphedInference
    L = _lib.lib()
    model = _small_detector(dev)
    head = model.bbox_head
    head.tower_streams = False
    det_flag = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    B, depth = 2, 3
    metas = [dict(img_shape=(256, 256, 3), pad_shape=(256, 256, 3), scale_factor=1.0, flip=False)] * B
    imgs = [torch.randn(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(20 + i)) for i in range(5)]
    target = [None]                                            # the buffer set the head's outputs are copied into (none: product path)

    def hook(m, inp, out):
        if target[0] is not None:
            flat = list(out[0]) + list(out[1]) + list(out[2])
            if not target[0]:
                target[0].extend(torch.empty_like(t) for t in flat)
            for b, t in zip(target[0], flat):
                b.copy_(t)
    handle = head.register_forward_hook(hook)
    assert L.orp_dcn_set_split_mode(mode) == 0
    try:
        def eager(img, store):
            target[0] = store
            with torch.no_grad():
                r = model.simple_test_batch(img, metas)
            target[0] = None
            return r
        want, want_det, store = [], [], []
        for im in imgs:
            want_det.append(eager(im, store))
            torch.cuda.synchronize()
            want.append([t.clone() for t in store])
        assert len({sum(len(c) for r in w for c in r) for w in want_det}) > 1
        slots, sets = [], []
        for d in range(depth):
            s_ = []
            eager(imgs[0], s_)                                 # (allocates this graph's buffers outside its capture)
            target[0] = s_
            slots.append(GraphedInference(model, imgs[0], metas)); sets.append(s_)
            target[0] = None
        streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
        from orientedreppoints_amd.mmdet_models.core import rbbox2result_packed
        rng = np.random.RandomState(3)
        for it in range(120):
            pick = [int(rng.randint(0, len(imgs))) for _ in range(depth)]
            cur = torch.cuda.current_stream(dev)
            for d in range(depth):
                streams[d].wait_stream(cur)
                with torch.cuda.stream(streams[d]):
                    slots[d].static_img.copy_(imgs[pick[d]], non_blocking=True)
                    slots[d].graph.replay()
            torch.cuda.synchronize()
            for d in range(depth):
                for a, b in zip(sets[d], want[pick[d]]):
                    assert torch.equal(a, b), "replay %d of graph %d: a head output differs from the eager step's" % (it, d)
                res = [rbbox2result_packed(p, head.num_classes) for p in slots[d].packed]
                for gr, wr in zip(res, want_det[pick[d]]):
                    assert gr is not None
                    for a, b in zip(gr, wr):
                        assert a.shape == b.shape and np.array_equal(a, b), "replay %d of graph %d: detections differ" % (it, d)
    finally:
        L.orp_dcn_set_split_mode(-1)
        handle.remove()
        torch.backends.cudnn.deterministic = det_flag


@pytest.mark.parametrize("mode", [6, 3])
def test_deformconv_pair_at_tile_height_one_next_to_other_streams_every_launch_compared(dev, mode):
    """The launch that produced round 4's wrong rows (small pyramid: tile height 1), 400 times next to a GEMM stream and a
    stream of tower convolutions, EVERY result compared on the device with the first one (which is checked against the exact
    fp32 kernel)."""
    from orientedreppoints_amd import _lib
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi
    L = _lib.lib()
    torch.manual_seed(5)
    sizes = (32, 16, 8, 4, 2)
    w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
    conv = nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
    side, third = torch.cuda.Stream(), torch.cuda.Stream()
    g = torch.randn(2048, 2048, device=dev)
    try:
        with torch.no_grad():
            for B in (1, 2):
                fa = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
                fb = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
                of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
                xa = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]

                def run():
                    r = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
                    return list(r[0]) + list(r[1])
                L.orp_dcn_set_split_mode(0)
                exact = [t.clone() for t in run()]
                L.orp_dcn_set_split_mode(mode)
                ref = [t.clone() for t in run()]
                for x, y in zip(ref, exact):
                    assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max())
                nbad = torch.zeros((), dtype=torch.int64, device=dev)
                for i in range(400):
                    if i % 4 == 0:
                        with torch.cuda.stream(side):
                            g = (g @ g).clamp_(-1, 1)
                    if i % 2 == 0:
                        with torch.cuda.stream(third):
                            conv_split_multi(xa, conv, xa, conv, nprod=6)
                    out = run()
                    nbad += torch.stack([(x != y).any() for x, y in zip(out, ref)]).any().to(torch.int64)
                torch.cuda.synchronize()
                assert int(nbad) == 0, "%d of 400 launches differ from the first one (B = %d)" % (int(nbad), B)
    finally:
        L.orp_dcn_set_split_mode(-1)
