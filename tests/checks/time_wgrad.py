"""Development aid (GPU box): the weight gradient of one 256 -> 256 3x3 tower layer over the five levels of B x 1024^2 images,
orp_conv_wgrad_split (one launch) against the library's kernel per level (torch.ops.aten.convolution_backward); torch events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_wgrad_split

dev = torch.device("cuda:0")
torch.manual_seed(0)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.02


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (1, 2):
    xs = [torch.randn(B, 256, 1024 // s, 1024 // s, device=dev) for s in (8, 16, 32, 64, 128)]
    gs = [torch.randn_like(x) * 0.01 for x in xs]
    flop = 2 * B * sum(x.size(2) * x.size(3) for x in xs) * 256 * 256 * 9
    ax = torch.tensor([max(float(x.abs().max()) for x in xs)], device=dev).view(torch.int32)
    ag = torch.tensor([max(float(g.abs().max()) for g in gs)], device=dev).view(torch.int32)
    t = timed(lambda: conv_wgrad_split(xs, gs, (256, 256, 3, 3)))
    t2 = timed(lambda: conv_wgrad_split(xs, gs, (256, 256, 3, 3), (1, 1), (1, 1), ax, ag))
    def lib():
        acc = None
        for x, g in zip(xs, gs):
            gw = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
            acc = gw if acc is None else acc + gw
        return acc
    tl = timed(lib)
    print("B=%d: orp_conv_wgrad_split %.1f us (own range pre-pass) / %.1f us (ranges handed in) = %.1f TFLOP/s fp32-equivalent; "
          "library, 5 launches + adds: %.1f us" % (B, t, t2, flop / t2 / 1e6, tl))
