"""Launch the hot-path kernels a few times on the bench shapes (for rocprofv3 --pmc passes; development aid).
usage: python tools/run_hot_kernels.py [dcn|dcn2|nms|bwd|ops|all] [iters]      (dcn2: the pair launch at 2 images = the tap-granular split; not part of `all`)"""
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
    from orientedreppoints_amd import _lib
    for mode in (-1, 0):          # the library's default arithmetic (bf16-split products), then the exact-fp32 MFMA kernel
        _lib.lib().orp_dcn_set_split_mode(mode)
        for _ in range(iters):
            deform_conv_forward_pair(xs, xs2, offs, w, w2, 1, 1, 1, relu=True)  # the head's launch: both layers, all levels
    _lib.lib().orp_dcn_set_split_mode(-1)
    # the towers' layer k of both towers as one launch (the same kernel without offsets), channels-last as in the head
    from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi
    ca, cb = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev), torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
    cl = [x.contiguous(memory_format=torch.channels_last) for x in xs]
    cl2 = [x.contiguous(memory_format=torch.channels_last) for x in xs2]
    with torch.no_grad():
        for _ in range(iters):
            conv_split_multi(cl, ca, cl2, cb)
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
if which in ('ops', 'all'):
    # the remaining hot-path kernels at the shapes bench.py's per_op_us table uses (configs[1] / configs[2])
    from orientedreppoints_amd.mmdet_ops import (ChamferDistance2D, box_iou_rotated, convex_giou, convex_iou, minaerarect,
                                                 points_in_quad_aligned, sigmoid_focal_loss)
    from orientedreppoints_amd.mmdet_ops.apaa import apaa_select, max_iou_assign, point_assign
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)   # noqa: E731
    IMG = 1024
    ar = []
    for st in (8, 16, 32, 64, 128):
        nn = IMG // st
        yy, xx = np.meshgrid(np.arange(nn), np.arange(nn), indexing='ij')
        ar.append(np.stack([xx.reshape(-1) * st + st / 2.0, yy.reshape(-1) * st + st / 2.0], 1))
    around = np.concatenate(ar)
    pall = t32(S.gen_pointsets(len(around), 6, around=around))
    g32 = t32(S.gen_gts(32, 3))
    p5344 = t32(S.gen_pointsets(5344, 2))
    P = 5000
    pp, gg = t32(S.gen_pointsets(P, 2)), t32(S.gen_gts(P, 3))
    rng = np.random.RandomState(0)
    ca, cb = t32(rng.rand(P, 40, 2) * 100), t32(rng.rand(P, 40, 2) * 100)
    N = 2 * 21824
    lg, lb = t32(rng.randn(N, 15)), torch.from_numpy(rng.randint(0, 16, N).astype(np.int64)).to(dev)
    pts3 = t32(np.concatenate([np.concatenate([a_ - st / 2.0, np.full((len(a_), 1), st)], 1) for a_, st in zip(ar, (8, 16, 32, 64, 128))]))
    g64 = t32(S.gen_gts(64, 5))
    ovl = t32((rng.rand(21824, 64) * 0.3) * (rng.rand(21824, 64) < 0.02))
    rb = t32(S.gen_rboxes(1000, 1))
    q = t32(rng.rand(P))
    pos_gt = torch.from_numpy(rng.randint(0, 64, P).astype(np.int64)).to(dev)
    pos_lvl = torch.from_numpy(rng.randint(0, 5, P).astype(np.int64)).to(dev)
    for _ in range(iters):
        convex_iou(pall, g32)
        convex_giou(pp, gg)
        minaerarect(p5344)
        ChamferDistance2D(ca, cb)
        points_in_quad_aligned(pp, gg)
        sigmoid_focal_loss(lg, lb, 2.0, 0.25)
        point_assign(pts3, g64)
        max_iou_assign(ovl, 0.1, 0.1)
        box_iou_rotated(rb, rb)
        apaa_select(q, pos_gt, pos_lvl, 64, 5)
torch.cuda.synchronize()
print('done')
