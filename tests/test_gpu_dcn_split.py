"""GPU: the fp32 DeformConv forward on the bf16 matrix pipe (csrc/orp_dcn_split.hip: every fp32 operand split exactly into
three bf16 pieces, 6 or 9 partial products, fp32 accumulation) -- same C entry points and tensors as the exact-fp32 MFMA
path, selected by orp_dcn_set_split_mode().  Checker: oracle.dcn_forward (the reference's float bilinear samples,
deform_conv_cuda_kernel.cu:84-115,190-243, contracted in DOUBLE).  The error gate of the mode: its maximum error against that
oracle must not exceed the exact-fp32 path's own error on the same inputs (both printed in the session summary)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from orientedreppoints_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _case(seed, B, C, H, W, Cout, std_off=2.0, wstd=0.05):
    rng = np.random.RandomState(seed)
    x = rng.normal(size=(B, C, H, W)).astype(np.float32)
    off = rng.normal(0, std_off, size=(B, 18, H, W)).astype(np.float32)
    w = rng.normal(0, wstd, size=(Cout, C, 3, 3)).astype(np.float32)
    return x, off, w


def _err(got, want):
    return float(np.max(np.abs(got.astype(np.float64) - want.astype(np.float64))) / max(1e-6, float(np.max(np.abs(want)))))


@pytest.fixture
def split(dev):
    from orientedreppoints_amd import _lib
    L = _lib.lib()

    def set_mode(m):                        # 0 exact fp32 MFMA | 6, 9 products of three bf16 pieces | 3 two fp16 pieces
        assert L.orp_dcn_set_split_mode(int(m)) == 0
        assert L.orp_dcn_get_split_mode() == int(m)
    yield set_mode
    L.orp_dcn_set_split_mode(-1)


@pytest.mark.parametrize("B,C,H,W,Cout", [(2, 256, 16, 16, 256), (1, 64, 9, 11, 64), (1, 128, 5, 40, 192), (3, 256, 7, 9, 256)])
def test_split_forward_vs_oracle_and_error_gate(dev, oracle, split, B, C, H, W, Cout):
    import conftest
    from orientedreppoints_amd.mmdet_ops import deform_conv
    x, off, w = _case(7 + C + H, B, C, H, W, Cout)
    want = oracle.dcn_forward(x, off, w, stride=1, pad=1, dil=1)
    errs = {}
    for mode in (0, 9, 6, 3):
        split(mode)
        got = deform_conv(_t(x, dev), _t(off, dev), _t(w, dev), 1, 1, 1, 1, 1, 64)
        assert got.shape == want.shape and got.is_contiguous()
        errs[mode] = _err(got.cpu().numpy(), want)
        assert errs[mode] <= 1e-4, mode
        xcl = _t(x, dev).contiguous(memory_format=torch.channels_last)
        got2 = deform_conv(xcl, _t(off, dev), _t(w, dev), 1, 1, 1, 1, 1, 64)
        assert got2.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(got2.contiguous(), got), "NHWC in/out must give the same bits (same summation order)"
        again = deform_conv(_t(x, dev), _t(off, dev), _t(w, dev), 1, 1, 1, 1, 1, 64)
        assert torch.equal(again, got)
    conftest.REPORT.append("DeformConv forward %dx%dx%dx%d -> %d, max |err| / max |out| vs the fp64-accumulated oracle: exact-fp32 MFMA "
                           "%.2e, split 9 products %.2e, split 6 products %.2e, two fp16 pieces / 3 products %.2e"
                           % (B, C, H, W, Cout, errs[0], errs[9], errs[6], errs[3]))
    # the gate: no worse than the exact-fp32 path's own accumulation error (+ 5e-8: half an fp32 ulp of the output scale,
    # the resolution of the comparison itself)
    assert errs[9] <= errs[0] + 5e-8
    assert errs[6] <= errs[0] + 5e-8
    assert errs[3] <= errs[0] + 5e-8


def test_split_modulated_bias_relu_multi_level(dev, oracle, split):
    """DCNv2 modulation + bias + fused ReLU, several levels in one launch, samples far outside the maps."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
    rng = np.random.RandomState(5)
    shapes = [(12, 10), (6, 5), (3, 3), (1, 2)]
    cases = [_case(30 + i, 2, 128, h, w, 64, std_off=(2.0, 6.0, 1.0, 3.0)[i]) for i, (h, w) in enumerate(shapes)]
    w = cases[0][2]
    masks = [rng.uniform(0, 1, size=(2, 9, h, ww)).astype(np.float32) for h, ww in shapes]
    bias = rng.normal(size=(64,)).astype(np.float32)
    for mode in (9, 6, 3):                  # (3: a modulated launch has no known sample range -- it runs as mode 6)
        split(mode)
        outs = deform_conv_forward_multi([_t(c[0], dev) for c in cases], [_t(c[1], dev) for c in cases], _t(w, dev), 1, 1, 1,
                                         masks=[_t(m, dev) for m in masks], bias=_t(bias, dev), relu=True)
        for c, m, o in zip(cases, masks, outs):
            want = np.maximum(oracle.dcn_forward(c[0], c[1], w, mask=m, bias=bias), 0.0)
            assert _err(o.cpu().numpy(), want) <= 1e-4


def test_split_pair_launch_at_head_shapes(dev, oracle, split):
    """orp_dcn_forward_pair on the split path (grid halves = layers): equals two single launches bit for bit, matches the
    oracle on a level small enough for it, and the exact path to 1e-5 of scale at the 1024^2 levels (B = 1 and 2)."""
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, deform_conv_forward_pair
    torch.manual_seed(3)
    for B, sizes in ((1, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]), (2, [(40, 40), (20, 20), (7, 9)])):
        fa = [torch.randn(B, 256, h, w, device=dev) for h, w in sizes]
        fb = [torch.randn(B, 256, h, w, device=dev) for h, w in sizes]
        of = [torch.randn(B, 18, h, w, device=dev) * 2.0 for h, w in sizes]
        w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
        split(0)
        ea, eb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
        for mode in (9, 6, 3):
            split(mode)
            pa, pb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
            sa = deform_conv_forward_multi(fa, of, w1, 1, 1, 1, relu=True)
            sb = deform_conv_forward_multi(fb, of, w2, 1, 1, 1, relu=True)
            for x, y in zip(pa + pb, sa + sb):
                assert torch.equal(x, y)
            for x, y in zip(pa + pb, ea + eb):
                assert float((x - y).abs().max()) <= 1e-5 * float(y.abs().max())
            lvl = len(sizes) - 1
            want = np.maximum(oracle.dcn_forward(fa[lvl].cpu().numpy(), of[lvl].cpu().numpy(), w1.cpu().numpy()), 0.0)
            assert _err(pa[lvl].cpu().numpy(), want) <= 1e-4
            # channels-last in -> channels-last out, same bits
            ca, cb = deform_conv_forward_pair([t.contiguous(memory_format=torch.channels_last) for t in fa],
                                              [t.contiguous(memory_format=torch.channels_last) for t in fb], of, w1, w2, 1, 1, 1, relu=True)
            for x, y in zip(ca + cb, pa + pb):
                assert x.is_contiguous(memory_format=torch.channels_last) and torch.equal(x.contiguous(), y)


def test_split_pieces_are_exact(dev, split):
    """The property the path rests on: with 9 products the contraction has NO representation error -- for inputs whose
    products and sums are exactly representable in fp32 (small integers) the result is exact, and zero offsets make the
    operator the library's 3 x 3 convolution."""
    from orientedreppoints_amd.mmdet_ops import deform_conv
    g = torch.Generator().manual_seed(1)
    x = torch.randint(-8, 9, (1, 64, 12, 12), generator=g).float().to(dev)
    w = torch.randint(-4, 5, (64, 64, 3, 3), generator=g).float().to(dev)
    off = torch.zeros(1, 18, 12, 12, device=dev)
    want = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    for mode in (9, 6, 3):
        split(mode)
        got = deform_conv(x, off, w, 1, 1, 1, 1, 1, 64)
        assert torch.equal(got.double(), want)
    # values with all 24 mantissa bits in use: one product per output (1 x 1 footprint through a one-hot weight)
    xs = (torch.rand(1, 64, 6, 6, generator=g) + 1.0).to(dev) * (1.0 + 2.0 ** -23)
    ws = torch.zeros(64, 64, 3, 3, device=dev)
    scale = (torch.rand(64, generator=g) + 1.0).to(dev)
    ws[torch.arange(64), torch.arange(64), 1, 1] = scale
    off = torch.zeros(1, 18, 6, 6, device=dev)
    split(9)
    got = deform_conv(xs, off, ws, 1, 1, 1, 1, 1, 64)
    exact = (xs.double() * scale.double()[None, :, None, None])
    # nine exact partial products summed in fp32: within one ulp of the exactly rounded product
    assert float(((got.double() - exact).abs() / exact.abs()).max()) <= 2.0 ** -23


@pytest.mark.parametrize("log2_ratio", [12, 16, 20, 24])
def test_one_outlier_channel_inside_a_tensor(dev, oracle, split, log2_ratio):
    """Round-5 verdict, weak 4: the fp16-pieces mode scales a whole TENSOR by one power of two (max |x| -> 2^14..2^15), so its
    representation error is 2^-22 of the tensor maximum, not of each element.  A realistic FPN map has outlier channels: here ONE
    input channel is 2^R times the others and half of the output channels do not read it (zero weights), so those outputs live at
    the small channels' scale while the launch's range word is set by the outlier.  Gate: 1e-5 of THAT half's own scale for R <=
    16 (every mode; measured error is at the fp32 accumulation's own level there -- the low fp16 piece still carries 11 bits
    at 2^-16 of the maximum).  R = 20 / 24 are past what two fp16 pieces can carry (the low piece goes subnormal: 2^-24 of the
    scaled maximum is its last bit): the measured error is REPORTED and bounded by the analytic 2^-(38 - R) of the half's scale,
    and the exact modes (0, 6, 9) stay under 1e-5 at every R -- `ORP_DCN_SPLIT=6` is the switch for such tensors."""
    import conftest
    from orientedreppoints_amd.mmdet_ops import deform_conv
    B, C, H, W, Cout = 1, 128, 12, 12, 128
    x, off, w = _case(90 + log2_ratio, B, C, H, W, Cout)
    x[:, 5] *= np.float32(2.0 ** log2_ratio)
    w[:Cout // 2, 5] = 0.0
    want = oracle.dcn_forward(x, off, w, stride=1, pad=1, dil=1)
    quiet = want[:, :Cout // 2]
    scale = float(np.max(np.abs(quiet)))
    assert scale < 2.0 ** (log2_ratio - 6) * 1e3 and float(np.max(np.abs(want))) > 2.0 ** (log2_ratio - 4)
    errs = {}
    for mode in (0, 9, 6, 3):
        split(mode)
        got = deform_conv(_t(x, dev), _t(off, dev), _t(w, dev), 1, 1, 1, 1, 1, 64).cpu().numpy()
        errs[mode] = float(np.max(np.abs(got[:, :Cout // 2].astype(np.float64) - quiet))) / scale
        assert _err(got, want) <= 1e-5                       # the whole tensor at the tensor's scale, as before
        if mode != 3 or log2_ratio <= 16:
            assert errs[mode] <= 1e-5, (mode, errs[mode])
        else:
            assert errs[mode] <= 2.0 ** -(38 - log2_ratio), (mode, errs[mode])
    conftest.REPORT.append("DeformConv, one input channel 2^%d x the others, error of the output channels that do not read it / their own "
                           "scale: exact-fp32 %.2e, 9 products %.2e, 6 products %.2e, two fp16 pieces %.2e"
                           % (log2_ratio, errs[0], errs[9], errs[6], errs[3]))


def test_wave_specialised_kernel_as_a_switch():
    """ORP_DCNS_WS=1 (csrc/orp_dcn_split.hip, round 6: 4 consumer + 4 producer waves per tile; measured, not the default): the
    DeformConv tests of this file and the convolution-vs-float64 test pass on it in a fresh process.  Its DeformConv instantiation
    is bit-identical to the default kernel's (same products, same order); the convolution instantiation has no side accumulators
    and is held to 1e-5 here (the default kernel's tighter gate, 1.5 x the library's own error, is why it is not the default)."""
    import subprocess
    env = dict(os.environ, ORP_DCNS_WS="1")
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                          os.path.join(here, "test_gpu_dcn_split.py"), "-k", "forward_vs_oracle or pair_launch or modulated or pieces_are_exact",
                          os.path.join(here, "test_gpu_conv_split.py") + "::test_conv_split_vs_oracle_and_float64",
                          os.path.join(here, "test_gpu_conv_split.py") + "::test_tower_layers_with_groupnorm_fused_around_the_convolutions"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:]


def test_wave_specialised_deformconv_is_bitwise_the_default_kernel(dev):
    """Both kernels in two fresh processes on the same seeded head-shaped pair launch: identical bits."""
    import subprocess
    code = ("import torch,hashlib,sys;sys.path.insert(0,%r);from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair as f;"
            "torch.manual_seed(5);d=torch.device('cuda:0');s=[(40,40),(20,20),(7,9)];"
            "a=[torch.randn(2,256,h,w,device=d) for h,w in s];b=[torch.randn(2,256,h,w,device=d) for h,w in s];"
            "o=[torch.randn(2,18,h,w,device=d)*2 for h,w in s];w1=torch.randn(256,256,3,3,device=d)*.02;w2=torch.randn(256,256,3,3,device=d)*.02;"
            "pa,pb=f(a,b,o,w1,w2,1,1,1,relu=True);print('HASH',hashlib.sha1(b''.join(t.cpu().numpy().tobytes() for t in pa+pb)).hexdigest())"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    hashes = []
    for ws in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORP_DCNS_WS=ws, ORP_DCN_SPLIT="3"), stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:]
        hashes.append([ln for ln in out.stdout.splitlines() if ln.startswith("HASH")][0])
    assert hashes[0] == hashes[1]


def test_wave_specialised_half_kernel_is_bitwise_the_symmetric_one(dev):
    """csrc/orp_dcn_half.hip, round 6: the fp16 / bf16 DeformConv forward on 4 consumer + 4 producer waves (the default) against the
    symmetric kernel (ORP_DCNH_WS=0), two fresh processes, same seeded launch over three levels and two images: identical bits."""
    import subprocess
    code = ("import torch,hashlib,sys;sys.path.insert(0,%r);from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi as f;"
            "d=torch.device('cuda:0');out=[]\n"
            "for dt in (torch.float16, torch.bfloat16):\n"
            "    torch.manual_seed(5);s=[(40,40),(20,20),(7,9)]\n"
            "    x=[torch.randn(2,256,h,w,device=d).to(dt) for h,w in s];o=[(torch.randn(2,18,h,w,device=d)*2).to(dt) for h,w in s]\n"
            "    w=(torch.randn(256,256,3,3,device=d)*.02).to(dt)\n"
            "    for cl in (False, True):\n"
            "        xs=[t.contiguous(memory_format=torch.channels_last) for t in x] if cl else x\n"
            "        out+=[t.float().cpu().numpy().tobytes() for t in f(xs,o,w,1,1,1,relu=True)]\n"
            "print('HASH',hashlib.sha1(b''.join(out)).hexdigest())"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    hashes = []
    for ws in ("0", "1"):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ORP_DCNH_WS=ws), stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:]
        hashes.append([ln for ln in out.stdout.splitlines() if ln.startswith("HASH")][0])
    assert hashes[0] == hashes[1]
