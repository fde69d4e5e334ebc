"""GPU (MI355X): the `double` branch of the reference's dispatch macros -- DeformConv / ModulatedDeformConv
(AT_DISPATCH_FLOATING_TYPES_AND_HALF, mmdet/ops/dcn/src/deform_conv_cuda_kernel.cu:259,353,451,720,777,838), sigmoid focal loss
(AT_DISPATCH_FLOATING_TYPES, sigmoid_focal_loss_cuda.cu:121,160) and pointsJf (points_justify_kernel.cu:107).  Rounds 1-5 narrowed
float64 tensors to fp32 silently (DeformConv) or refused them; round 6: float64 tensors are computed in float64 -- DeformConv as the
reference's own column formulation (HIP sampling kernels templated on double around the library's double GEMM), focal as the
double instantiation of the reference's template, pointsJf exactly as its double instantiation behaves (a float point struct)."""
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


def _dcn_ref_f64(x, off, w, mask=None, bias=None, stride=1, pad=1, dil=1, groups=1, dg=1):
    """Plain numpy float64 restatement of deformable_im2col + the grouped contraction (deform_conv_cuda_kernel.cu:84-115,190-243)."""
    B, C, H, W = x.shape
    Cout, Cg, kh, kw = w.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    col = np.zeros((B, C, kh * kw, Ho, Wo))
    cpdg = C // dg
    for b in range(B):
        for c in range(C):
            g = c // cpdg
            for t in range(kh * kw):
                ki, kj = divmod(t, kw)
                for ho in range(Ho):
                    for wo in range(Wo):
                        h = ho * stride - pad + ki * dil + off[b, (g * kh * kw + t) * 2, ho, wo]
                        ww = wo * stride - pad + kj * dil + off[b, (g * kh * kw + t) * 2 + 1, ho, wo]
                        if not (h > -1 and ww > -1 and h < H and ww < W):
                            continue
                        hl, wl = int(np.floor(h)), int(np.floor(ww))
                        lh, lw = h - hl, ww - wl
                        v = 0.0
                        for (hh, wq, cf) in ((hl, wl, (1 - lh) * (1 - lw)), (hl, wl + 1, (1 - lh) * lw),
                                             (hl + 1, wl, lh * (1 - lw)), (hl + 1, wl + 1, lh * lw)):
                            if 0 <= hh <= H - 1 and 0 <= wq <= W - 1:
                                v += cf * x[b, c, hh, wq]
                        if mask is not None:
                            v *= mask[b, g * kh * kw + t, ho, wo]
                        col[b, c, t, ho, wo] = v
    out = np.zeros((B, Cout, Ho, Wo))
    og = Cout // groups
    for g in range(groups):
        wg = w[g * og:(g + 1) * og].reshape(og, -1)
        cg = col[:, g * Cg:(g + 1) * Cg].reshape(B, Cg * kh * kw, Ho * Wo)
        out[:, g * og:(g + 1) * og] = np.einsum('ok,bkp->bop', wg, cg).reshape(B, og, Ho, Wo)
    if bias is not None:
        out += bias[None, :, None, None]
    return out


def _offsets(rng, B, taps2, Ho, Wo):
    """Offsets whose sample points stay 0.15 away from integer coordinates (the bilinear kink), so that gradcheck's central
    differences see a smooth function."""
    base = rng.randint(-2, 3, size=(B, taps2, Ho, Wo)).astype(np.float64)
    return base + rng.uniform(0.15, 0.85, size=base.shape)


@pytest.mark.parametrize("groups,dg", [(1, 1), (2, 2)])
def test_deform_conv_float64_forward_and_gradcheck(dev, groups, dg):
    from orientedreppoints_amd.mmdet_ops.deform_conv import DeformConvFunction
    rng = np.random.RandomState(3 + groups)
    B, C, H, W, Cout = 2, 4, 5, 6, 4
    x = rng.normal(size=(B, C, H, W))
    off = _offsets(rng, B, 18 * dg, H, W)
    w = rng.normal(size=(Cout, C // groups, 3, 3)) * 0.3
    tx, toff, tw = (torch.from_numpy(a).to(dev).requires_grad_(True) for a in (x, off, w))
    out = DeformConvFunction.apply(tx, toff, tw, 1, 1, 1, groups, dg, 64)
    assert out.dtype == torch.float64
    want = _dcn_ref_f64(x, off, w, groups=groups, dg=dg)
    assert np.max(np.abs(out.detach().cpu().numpy() - want)) <= 1e-12 * max(1.0, np.abs(want).max())      # DOUBLE, not narrowed to fp32
    # fp32 tensors of the same values differ at fp32 level: the test above could not pass on a narrowed path
    out32 = DeformConvFunction.apply(tx.detach().float(), toff.detach().float(), tw.detach().float(), 1, 1, 1, groups, dg, 64)
    assert 1e-9 < float((out32.double() - out.detach()).abs().max()) < 1e-4
    assert torch.autograd.gradcheck(lambda a, b, c: DeformConvFunction.apply(a, b, c, 1, 1, 1, groups, dg, 64), (tx, toff, tw),
                                    eps=1e-6, atol=1e-6, rtol=1e-5, nondet_tol=1e-9)


def test_modulated_deform_conv_float64_forward_and_gradcheck(dev):
    from orientedreppoints_amd.mmdet_ops.deform_conv import ModulatedDeformConvFunction
    rng = np.random.RandomState(9)
    B, C, H, W, Cout = 1, 4, 5, 5, 6
    x = rng.normal(size=(B, C, H, W))
    off = _offsets(rng, B, 18, H, W)
    m = rng.uniform(0.1, 0.9, size=(B, 9, H, W))
    w = rng.normal(size=(Cout, C, 3, 3)) * 0.3
    bias = rng.normal(size=(Cout,))
    ts = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (x, off, m, w, bias)]
    out = ModulatedDeformConvFunction.apply(*ts, 1, 1, 1, 1, 1)
    assert out.dtype == torch.float64
    want = _dcn_ref_f64(x, off, w, mask=m, bias=bias)
    assert np.max(np.abs(out.detach().cpu().numpy() - want)) <= 1e-12 * max(1.0, np.abs(want).max())
    assert torch.autograd.gradcheck(lambda *a: ModulatedDeformConvFunction.apply(*a, 1, 1, 1, 1, 1), tuple(ts),
                                    eps=1e-6, atol=1e-6, rtol=1e-5, nondet_tol=1e-9)


def test_modules_take_float64(dev):
    """DeformConvPack / ModulatedDeformConvPack moved to double (`.double()`) compute in double end to end."""
    from orientedreppoints_amd.mmdet_ops import DeformConvPack, ModulatedDeformConvPack
    torch.manual_seed(0)
    for cls in (DeformConvPack, ModulatedDeformConvPack):
        mod = cls(8, 8, 3, padding=1).to(dev).double()
        torch.nn.init.normal_(mod.conv_offset.weight, 0, 0.1)
        x = torch.randn(1, 8, 6, 6, device=dev, dtype=torch.float64, requires_grad=True)
        y = mod(x)
        assert y.dtype == torch.float64
        y.square().sum().backward()
        assert x.grad.dtype == torch.float64 and mod.weight.grad.dtype == torch.float64
        y32 = cls(8, 8, 3, padding=1).to(dev)
        y32.load_state_dict({k: v.float() for k, v in mod.state_dict().items()})
        with torch.no_grad():
            d = float((y32(x.detach().float()).double() - y.detach()).abs().max())
        assert 0 < d < 1e-4


def test_sigmoid_focal_loss_float64(dev):
    from orientedreppoints_amd.mmdet_ops.sigmoid_focal_loss import sigmoid_focal_loss_cuda as ext
    rng = np.random.RandomState(1)
    n, c = 300, 15
    x = rng.normal(0, 3, size=(n, c))
    t = rng.randint(0, c + 1, size=n).astype(np.int64)
    gamma, alpha = 2.0, 0.25
    tx, tt = torch.from_numpy(x).to(dev), torch.from_numpy(t).to(dev)
    loss = ext.forward(tx, tt, c, gamma, alpha)
    assert loss.dtype == torch.float64
    p = 1.0 / (1.0 + np.exp(-x))
    pos = (t[:, None] == np.arange(1, c + 1)[None, :]).astype(np.float64)
    neg = (t[:, None] >= 0).astype(np.float64) * (1.0 - pos)
    want = -pos * (1 - p) ** gamma * np.log(np.maximum(p, np.finfo(np.float32).tiny)) * alpha \
        - neg * p ** gamma * (-x * (x >= 0) - np.log1p(np.exp(x - 2 * x * (x >= 0)))) * (1 - alpha)
    got = loss.cpu().numpy()
    assert np.max(np.abs(got - want)) <= 2e-6 * max(1.0, np.abs(want).max())          # its expf / logf / powf are single precision, as the reference's
    l32 = ext.forward(tx.float(), tt, c, gamma, alpha)
    assert float((l32.double() - loss).abs().max()) <= 1e-5
    d = torch.from_numpy(rng.uniform(0.5, 1.5, size=(n, c))).to(dev)
    g = ext.backward(tx, tt, d, c, gamma, alpha)
    assert g.dtype == torch.float64
    g32 = ext.backward(tx.float(), tt, d.float(), c, gamma, alpha)
    assert float((g32.double() - g).abs().max()) <= 1e-5
    # the gradient's closed form (sigmoid_focal_loss_cuda.cu:73-96) in numpy float64
    logp = np.log(np.maximum(p, np.finfo(np.float32).tiny))
    log1mp = -x * (x >= 0) - np.log1p(np.exp(x - 2 * x * (x >= 0)))
    want_g = (-pos * (1 - p) ** gamma * (1 - p - p * gamma * logp) * alpha
              - neg * p ** gamma * (log1mp * (1 - p) * gamma - p) * (1 - alpha)) * d.cpu().numpy()
    assert np.max(np.abs(g.cpu().numpy() - want_g)) <= 2e-6 * max(1.0, np.abs(want_g).max())


def test_points_justify_float64_is_the_reference_double_instantiation(dev):
    from orientedreppoints_amd import synthetic as S
    from orientedreppoints_amd.mmdet_ops import pointsJf
    quads = S.gen_gts(40, 5)
    pts = S.gen_pointsets(30, 6, around=quads.reshape(-1, 4, 2).mean(1)[:30]).reshape(-1, 2)
    p64, q64 = torch.from_numpy(pts).to(dev), torch.from_numpy(quads).to(dev)
    out64 = torch.full((p64.size(0), q64.size(0)), -1.0, device=dev, dtype=torch.float64)
    assert pointsJf(p64, q64, out64) == 1
    out32 = torch.full((p64.size(0), q64.size(0)), -1.0, device=dev)
    assert pointsJf(p64.float(), q64.float(), out32) == 1
    assert torch.equal(out64, out32.double()) and 0 < float(out64.sum()) < out64.numel()
    with pytest.raises(TypeError):
        pointsJf(p64, q64.float(), out64)
