"""Small-level 3x3 convolution: one HIP launch for the 32^2/16^2/8^2 levels vs the framework's per-level calls (dev aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from orientedreppoints_amd.mmdet_ops import fused_norm as FN
dev = torch.device("cuda:0")
torch.manual_seed(0)
conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B in (1, 2):
    xs = [torch.randn(B, 256, h, h, device=dev) for h in (32, 16, 8)]
    with torch.no_grad():
        t_hip = timeit(lambda: FN.conv3x3_multi(xs, conv))
        t_lib = timeit(lambda: [F.conv2d(x, conv.weight, None, padding=1) for x in xs])
        t_l2 = timeit(lambda: F.conv2d(xs[0], conv.weight, None, padding=1))
    print("B=%d small levels (32,16,8): HIP one launch %.1f us   library 3 calls %.1f us (32^2 alone %.1f us)" % (B, t_hip, t_lib, t_l2))
