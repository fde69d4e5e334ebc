"""Development aid (GPU box): does the per-workgroup time of the DeformConv pair kernel depend on how many CUs are busy?
Whole-tile launches (ORP_DCN_KSPLIT=0) of level sets with 171 ... 254 tiles: every workgroup does the same 18 tap steps,
so the launch time IS the per-workgroup time."""
import os, sys
os.environ['ORP_DCN_KSPLIT'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair

dev = torch.device("cuda:0")
torch.manual_seed(0)
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
for sizes in ((64,), (128,), (128, 64), (128, 64, 32, 16, 8), (128, 64, 32, 32, 16, 8), (128, 64, 32, 32, 32, 16, 8), (128, 64, 48, 32, 16)):
    tiles = sum((n * n + 95) // 96 for n in sizes)
    fa = [torch.randn(1, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
    fb = [torch.randn(1, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
    of = [torch.randn(1, 18, n, n, device=dev) * 2 for n in sizes]
    for _ in range(5):
        deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40):
        deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    e1.record(); torch.cuda.synchronize()
    print("levels %-28s tiles %3d: %.1f us per launch" % (sizes, tiles, e0.elapsed_time(e1) / 40 * 1e3))
