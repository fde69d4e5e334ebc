"""Development aid (GPU box): time the pair convolution and the DeformConv pair launch at the BASELINE shapes with whatever library
ORP_HIP_LIB names -- the ORP_DCNS_DBG timing variants of tools/build_variant.py (wrong results by construction) give the anatomy of a phase."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.dota_configs import r50_model
from orientedreppoints_amd.mmdet_models import ConfigDict
from orientedreppoints_amd.mmdet_models.registry import build_head
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi, to_channels_last_multi

dev = torch.device("cuda:0")
torch.manual_seed(0)
head = build_head(ConfigDict(r50_model['bbox_head'])).to(dev).eval()


def timed(fn, n=40):
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
    B, size = int(os.environ.get("BATCH", "1")), int(os.environ.get("SIZE", "1024"))
    feats = [torch.randn(B, 256, size // s, size // s, device=dev) for s in (8, 16, 32, 64, 128)]
    cl = to_channels_last_multi(feats)
    a, b = head.cls_convs[0], head.reg_convs[0]
    out = []
    for nprod in (3, 6):
        out.append("conv pair %d products %.1f us" % (nprod, timed(lambda: conv_split_multi(cl, a.conv, cl, b.conv, nprod=nprod))))
    out.append("conv single 3 products %.1f us" % timed(lambda: conv_split_multi(cl, a.conv, nprod=3)))
    print(os.environ.get("ORP_HIP_LIB", "in-tree").split("liborp_hip_")[-1], "|", " | ".join(out))
