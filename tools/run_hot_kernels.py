"""Launch the hot-path kernels a few times on the bench shapes (for rocprofv3 --pmc passes; development aid).
usage: python tools/run_hot_kernels.py [dcn|dcn2|nms|bwd|all] [iters]      (dcn2: the pair launch at 2 images = the tap-granular split; not part of `all`)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda:0')
if which in ('dcn', 'all'):
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, deform_conv_forward_pair
    torch.manual_seed(0)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    xs = [torch.randn(1, 256, h, h, device=dev) for h in (128, 64, 32, 16, 8)]      # NCHW, as the towers hand them over
    xs2 = [torch.randn_like(x) for x in xs]
    offs = [torch.randn(1, 18, h, h, device=dev) * 2 for h in (128, 64, 32, 16, 8)]
    for _ in range(iters):
        deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)      # the head's launch: both layers, all levels
    hx, ho, hw = [x.half() for x in xs], [o.half() for o in offs], w.half()
    for _ in range(iters):
        deform_conv_forward_multi(hx, ho, hw, 1, 1, 1)                          # fp16 path
if which == 'dcn2':
    from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
    torch.manual_seed(0)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    xs = [torch.randn(2, 256, h, h, device=dev) for h in (128, 64, 32, 16, 8)]
    xs2 = [torch.randn_like(x) for x in xs]
    offs = [torch.randn(2, 18, h, h, device=dev) * 2 for h in (128, 64, 32, 16, 8)]
    for _ in range(iters):
        deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)
if which in ('nms', 'all'):
    from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
    d, _ = S.gen_dense_scene(2000, 1)
    t = torch.from_numpy(d.astype(np.float32)).to(dev)
    for _ in range(iters):
        rnms_device(t, 0.4)
if which in ('bwd', 'all'):
    from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw
    torch.manual_seed(1)
    B = 2
    xs = [torch.randn(B, 256, h, h, device=dev) for h in (128, 64, 32, 16, 8)]
    offs = [torch.randn(B, 18, h, h, device=dev) * 2 for h in (128, 64, 32, 16, 8)]
    gos = [torch.randn(B, 256, h, h, device=dev) for h in (128, 64, 32, 16, 8)]
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.01
    for _ in range(iters):
        bw.backward_mfma(xs, offs, w, gos, (1, 1), (1, 1), (1, 1))         # dense gradients, 2 x 21 824 positions
torch.cuda.synchronize()
print('done')
