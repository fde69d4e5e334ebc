"""DCN forward timing on the bench shapes (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
torch.manual_seed(0)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
for B in (1, 2):
    xs = [torch.randn(B, 256, h, h, device=dev) for h in (128, 64, 32, 16, 8)]
    offs = [torch.randn(B, 18, h, h, device=dev) * 2 for h in (128, 64, 32, 16, 8)]
    fl = 2 * 21824 * B * 256 * 2304
    xcl = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    us = timeit(lambda: deform_conv_forward_multi(xcl, offs, w, 1, 1, 1), iters=10)
    ref = deform_conv_forward_multi(xs, offs, w, 1, 1, 1)
    got = deform_conv_forward_multi(xcl, offs, w, 1, 1, 1)
    err = max(float((a - b).abs().max()) for a, b in zip(ref, got))
    print("DCN fwd all levels B=%d NHWC: %.1f us  %.1f TFLOP/s  (nchw-vs-nhwc max abs diff %.2e)" % (B, us, fl / us / 1e6, err))
