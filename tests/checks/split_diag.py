"""Development aid (GPU box): pair vs single launches on the split path -- where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, deform_conv_forward_pair
dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(3)
for B, sizes in ((1, [(128, 128), (64, 64), (32, 32), (16, 16), (8, 8)]), (2, [(40, 40), (20, 20), (7, 9)]), (1, [(16, 16)])):
    fa = [torch.randn(B, 256, h, w, device=dev) for h, w in sizes]
    fb = [torch.randn(B, 256, h, w, device=dev) for h, w in sizes]
    of = [torch.randn(B, 18, h, w, device=dev) * 2.0 for h, w in sizes]
    w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
    for mode in (9, 6):
        L.orp_dcn_set_split_mode(mode)
        pa, pb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
        pa2, pb2 = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
        sa = deform_conv_forward_multi(fa, of, w1, 1, 1, 1, relu=False)
        sb = deform_conv_forward_multi(fb, of, w2, 1, 1, 1, relu=False)
        sa2 = deform_conv_forward_multi(fa, of, w1, 1, 1, 1, relu=False)
        for name, xs, ys in (("pairA-vs-singleA", pa, sa), ("pairB-vs-singleB", pb, sb), ("pairA-vs-pairA", pa, pa2), ("singleA-vs-singleA", sa, sa2)):
            for lvl, (x, y) in enumerate(zip(xs, ys)):
                d = (x - y).abs()
                nbad = int((d > 0).sum())
                if nbad:
                    idx = torch.nonzero(d > 0)
                    print("B=%d mode %d %s level %d: %d of %d differ, max %.3e (scale %.2f), nan %d; first idx %s; channels %s positions(h) %s"
                          % (B, mode, name, lvl, nbad, d.numel(), float(d.max()), float(y.abs().max()), int(torch.isnan(x).sum()), idx[0].tolist(),
                             sorted(set(idx[:, 1].tolist()))[:8], sorted(set(idx[:, 2].tolist()))[:8]))
print("done")
