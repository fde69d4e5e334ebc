"""Development aid (GPU box): does the library's own fused convolution + bias + ReLU (aten::miopen_convolution_relu /
miopen_convolution_add_relu, MIOpen fusion plans) beat convolution + this repo's one-pass affine (orp_affine_act) for the backbone's
eval-mode BatchNorm?  The backbone stays on the stock library either way; the question is only whether the normalisation pass
(affine_act_kernel: 53 launches, ~10 % of the inference step) can ride in the library's epilogue with the scale folded into the weights.
    python tests/checks/time_backbone_fusion.py
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from orientedreppoints_amd import _lib  # noqa: E402


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6


def affine(x, res, scale, shift, relu):
    rc = _lib.lib().orp_affine_act(_lib.ptr(x), _lib.ptr(res), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(x), x.size(0), x.size(1),
                                   x.size(2) * x.size(3), 1 if relu else 0, _lib.stream_of(x))
    _lib.check(rc, "orp_affine_act")
    return x


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    # (name, Cin, Cout, k, stride, H_in, residual)   R-50 at 1024^2, one image
    shapes = [
        ('stem 7x7/2', 3, 64, 7, 2, 1024, False),
        ('l1 conv1 1x1 (256->64)', 256, 64, 1, 1, 256, False),
        ('l1 conv2 3x3 (64)', 64, 64, 3, 1, 256, False),
        ('l1 conv3 1x1 (64->256) +res', 64, 256, 1, 1, 256, True),
        ('l2 conv1 1x1 (512->128)', 512, 128, 1, 1, 128, False),
        ('l2 conv2 3x3 (128)', 128, 128, 3, 1, 128, False),
        ('l2 conv2 3x3/2 (128)', 128, 128, 3, 2, 256, False),
        ('l2 conv3 1x1 (128->512) +res', 128, 512, 1, 1, 128, True),
        ('l3 conv1 1x1 (1024->256)', 1024, 256, 1, 1, 64, False),
        ('l3 conv2 3x3 (256)', 256, 256, 3, 1, 64, False),
        ('l3 conv3 1x1 (256->1024) +res', 256, 1024, 1, 1, 64, True),
        ('l4 conv1 1x1 (2048->512)', 2048, 512, 1, 1, 32, False),
        ('l4 conv2 3x3 (512)', 512, 512, 3, 1, 32, False),
        ('l4 conv3 1x1 (512->2048) +res', 512, 2048, 1, 1, 32, True),
        ('l2 downsample 1x1/2 (256->512)', 256, 512, 1, 2, 256, False),
    ]
    tot_now = tot_fused = 0.0
    for name, cin, cout, k, s, h, has_res in shapes:
        x = torch.randn(1, cin, h, h, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * (1.0 / (cin * k * k) ** 0.5)
        scale = torch.rand(cout, device=dev) + 0.5
        shift = torch.randn(cout, device=dev) * 0.1
        pad = k // 2
        ho = (h + 2 * pad - k) // s + 1
        res = torch.randn(1, cout, ho, ho, device=dev) if has_res else None
        wf = (w * scale.view(-1, 1, 1, 1)).contiguous()
        relu = True

        def now():
            return affine(F.conv2d(x, w, None, s, pad), res, scale, shift, relu)

        def fused():
            if has_res:
                return torch.ops.aten.miopen_convolution_add_relu(x, wf, res, 1.0, shift, [s, s], [pad, pad], [1, 1], 1)
            return torch.ops.aten.miopen_convolution_relu(x, wf, shift, [s, s], [pad, pad], [1, 1], 1)

        def conv_only():
            return F.conv2d(x, w, None, s, pad)

        ref = now().clone()
        try:
            out = fused()
            err = ((out - ref).abs().max() / ref.abs().max()).item()
            t_f = timed(fused)
        except Exception as e:   # noqa: BLE001
            err, t_f = float('nan'), float('nan')
            print('   fused op failed:', str(e).splitlines()[0][:160])
        t_n, t_c = timed(now), timed(conv_only)
        same = torch.equal(fused(), fused()) if t_f == t_f else None
        print('%-34s conv %7.1f us | conv + affine pass %7.1f us | library fused %7.1f us | max err / scale %.2e | fused reproducible %s'
              % (name, t_c, t_n, t_f, err, same))
        tot_now += t_n
        tot_fused += t_f


if __name__ == '__main__':
    main()
