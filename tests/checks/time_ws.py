"""Development aid (GPU box): the fp16-pieces split kernel at the BASELINE shapes -- the head's DeformConv pair launch and the towers'
PLAIN pair convolution, 1024^2 B = 1, channels-last in / out (no layout passes in the timed region) -- timed with HIP events around
back-to-back launches.  ORP_DCNS_WS=0|1 selects the symmetric / the wave-specialised kernel; ORP_HIP_LIB a variant build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.dota_configs import r50_model
from orientedreppoints_amd.mmdet_models import ConfigDict
from orientedreppoints_amd.mmdet_models.registry import build_head
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi, to_channels_last_multi

dev = torch.device("cuda:0")
torch.manual_seed(0)
head = build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()
tag = "%s WS=%s" % (os.path.basename(os.environ.get("ORP_HIP_LIB", "in-tree")), os.environ.get("ORP_DCNS_WS", "default"))


def timed(fn, n=40):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for size, B in ((1024, 1), (1536, 1)):
        sizes = [size // s for s in (8, 16, 32, 64, 128)]
        feats = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
        cl = to_channels_last_multi(feats)
        a = [x.contiguous(memory_format=torch.channels_last) for x in feats]
        of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
        w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
        flop = 2 * 2 * B * sum(s * s for s in sizes) * 256 * 256 * 9
        t = timed(lambda: deform_conv_forward_pair(a, a, of, w1, w2, 1, 1, 1, relu=True))
        print("[%s] %d^2 DeformConv pair (incl. range pre-pass): %.1f us = %.0f TFLOP/s issued" % (tag, size, t, 3 * flop / t / 1e6))
        t = timed(lambda: conv_split_multi(cl, head.cls_convs[0].conv, cl, head.reg_convs[0].conv, nprod=3))
        print("[%s] %d^2 PLAIN pair convolution (incl. range pre-pass): %.1f us = %.0f TFLOP/s issued" % (tag, size, t, 3 * flop / t / 1e6))
