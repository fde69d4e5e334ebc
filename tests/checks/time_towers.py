"""Development aid (GPU box): the head's towers at the BASELINE shapes (1024^2 -> 128^2 .. 8^2 levels), library convolutions
+ small-level kernel + NCHW GroupNorm (split_towers = False) against the channels-last path on the bf16 matrix pipe
(orp_conv_split_multi + orp_groupnorm_act_multi_cl), and the pieces of the latter; torch events around back-to-back calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.dota_configs import r50_model
from orientedreppoints_amd.mmdet_models import ConfigDict
from orientedreppoints_amd.mmdet_models.registry import build_head
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi, group_norm_act_multi_cl, to_channels_last_multi

dev = torch.device("cuda:0")
torch.manual_seed(0)
head = build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()


def timed(fn, n=30):
    for _ in range(5):
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
    for size, B in ((1024, 1), (1024, 2), (1536, 1)):
        feats = [torch.randn(B, 256, size // s, size // s, device=dev) for s in (8, 16, 32, 64, 128)]
        head.tower_streams = False
        for flag in (False, True):
            head.split_towers = flag
            print("%d^2 B=%d head forward (towers + DeformConv pair + output convolutions), split_towers=%s: %.1f us"
                  % (size, B, flag, timed(lambda: head(feats))))
        cl = to_channels_last_multi(feats)
        a, b = head.cls_convs[0], head.reg_convs[0]
        n = len(feats)
        flop = 2 * B * sum(f.size(2) * f.size(3) for f in feats) * 256 * 256 * 9
        t = timed(lambda: to_channels_last_multi(feats))
        print("   to_channels_last_multi: %.1f us" % t)
        for nprod in (3, 6, 9):
            t = timed(lambda: conv_split_multi(cl, a.conv, cl, b.conv, nprod=nprod))
            print("   pair convolution, %d products: %.1f us = %.1f TFLOP/s fp32-equivalent (%.0f TFLOP/s bf16 issued)"
                  % (nprod, t, 2 * flop / t / 1e6, 2 * flop * nprod / t / 1e6))
            t = timed(lambda: conv_split_multi(cl, a.conv, nprod=nprod))
            print("   single convolution, %d products: %.1f us = %.1f TFLOP/s fp32-equivalent" % (nprod, t, flop / t / 1e6))
        t = timed(lambda: conv_split_multi(cl, head.reppoints_pts_init_conv, bias=True, relu=True, out_channels_last=False))
        print("   single convolution + bias + ReLU, NCHW out: %.1f us" % t)
        oa, ob = conv_split_multi(cl, a.conv, cl, b.conv)
        t = timed(lambda: group_norm_act_multi_cl(oa + ob, [a.norm] * n + [b.norm] * n, relu=True, inplace=False))
        print("   GroupNorm+ReLU channels-last, both towers (10 tensors): %.1f us" % t)
