"""Quick per-op timing on the GPU box (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S, _lib
from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
from orientedreppoints_amd.mmdet_ops import minaerarect, convex_iou

dev = torch.device("cuda:0")

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us

for n in (500, 2000, 5000, 16000):
    d, _ = S.gen_dense_scene(n, 1)
    t = torch.from_numpy(d.astype(np.float32)).to(dev)
    us = timeit(lambda: rnms_device(t, 0.4), iters=10)
    keep, num = rnms_device(t, 0.4)
    print("rnms dense-scene n=%d: %.1f us  kept=%d  pairs=%.2fM  %.1f ns/pair-upper" % (n, us, int(num.item()), n*n/2e6, us*1e3/(n*n/2)))
for n in (2000,):
    d = S.gen_polys(n, 1, clustered=True)
    t = torch.from_numpy(d.astype(np.float32)).to(dev)
    us = timeit(lambda: rnms_device(t, 0.4), iters=10)
    print("rnms clustered n=%d: %.1f us" % (n, us))
pts = torch.from_numpy(S.gen_pointsets(5344, 2).astype(np.float32)).to(dev)
print("minaerarect 5344: %.1f us" % timeit(lambda: minaerarect(pts)))
for k in (8, 64, 256):
    gts = torch.from_numpy(S.gen_gts(k, 3).astype(np.float32)).to(dev)
    p = torch.from_numpy(S.gen_pointsets(21824, 4).astype(np.float32)).to(dev)
    us = timeit(lambda: convex_iou(p, gts), iters=5)
    print("convex_iou 21824 x %d: %.1f us (%.1f ns/pair)" % (k, us, us*1e3/(21824*k)))
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
for B in (1, 2):
    xs = [torch.randn(B, 256, h, h, device=dev) for h in (128, 64, 32, 16, 8)]
    offs = [torch.randn(B, 18, h, h, device=dev) * 2 for h in (128, 64, 32, 16, 8)]
    us = timeit(lambda: deform_conv_forward_multi(xs, offs, w, 1, 1, 1), iters=10)
    fl = 2 * 21824 * B * 256 * 2304
    print("DCN fwd all levels B=%d NCHW: %.1f us  %.1f TFLOP/s" % (B, us, fl / us / 1e6))
    xcl = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    us = timeit(lambda: deform_conv_forward_multi(xcl, offs, w, 1, 1, 1), iters=10)
    print("DCN fwd all levels B=%d NHWC: %.1f us  %.1f TFLOP/s" % (B, us, fl / us / 1e6))
