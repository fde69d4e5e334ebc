"""Development aid (GPU box): split path vs exact path at multi-round shapes -- where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, deform_conv_forward_pair
dev = torch.device("cuda:0")
L = _lib.lib()
torch.manual_seed(3)
for B, img in ((2, 1024), (3, 1024), (1, 1536)):
    sizes = [img // s for s in (8, 16, 32, 64, 128)]
    fa = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    fb = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    of = [torch.randn(B, 18, n, n, device=dev) * 2.0 for n in sizes]
    w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
    L.orp_dcn_set_split_mode(0)
    ea, eb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
    es = deform_conv_forward_multi(fa, of, w1, 1, 1, 1, relu=False)
    for mt in (os.environ.get("MTS", "0").split(",")):
        L.orp_dcn_set_split_mode(6)
        pa, pb = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
        sa = deform_conv_forward_multi(fa, of, w1, 1, 1, 1, relu=False)
        for name, xs, ys in (("pairA", pa, ea), ("pairB", pb, eb), ("singleA", sa, es)):
            for lvl, (x, y) in enumerate(zip(xs, ys)):
                d = (x - y).abs()
                tol = 1e-5 * float(y.abs().max())
                nbad = int((d > tol).sum())
                if nbad:
                    idx = torch.nonzero(d > tol)
                    print("B=%d img %d %s level %d: %d of %d beyond 1e-5, max %.3e (scale %.2f), nan %d; first %s last %s; batch %s"
                          % (B, img, name, lvl, nbad, d.numel(), float(d.max()), float(y.abs().max()), int(torch.isnan(x).sum()), idx[0].tolist(), idx[-1].tolist(),
                             sorted(set(idx[:, 0].tolist()))))
print("done")
