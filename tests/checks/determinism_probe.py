"""Development aid (GPU box): which stage of the eager inference step is not bitwise reproducible run to run?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as Bn
from orientedreppoints_amd.dota_configs import test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_detector(ConfigDict(Bn.MODELS['r50']), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
img = torch.randn(1, 3, 1024, 1024, device=dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)]
Bn.calibrate_head(model, img)


def stages():
    with torch.no_grad():
        c = model.backbone(img)
        f = model.neck(c)
        outs = model.bbox_head(f)
    return dict(backbone=[t.clone() for t in c], neck=[t.clone() for t in f],
                cls=[t.clone() for t in outs[0]], pts_init=[t.clone() for t in outs[1]], pts_refine=[t.clone() for t in outs[2]])


ref = stages()
for rep in range(4):
    cur = stages()
    msg = []
    for k in ref:
        diffs = [float((a - b).abs().max()) for a, b in zip(ref[k], cur[k])]
        msg.append("%s %s" % (k, ["%.1e" % d for d in diffs]))
    print("run %d vs run 0: " % (rep + 1) + " | ".join(msg))
# the head alone on FIXED features
f = [t.clone() for t in ref['neck']]
with torch.no_grad():
    h0 = model.bbox_head(f)
    for rep in range(3):
        h1 = model.bbox_head(f)
        print("head on fixed features, run %d: max diffs cls %s pts_init %s pts_refine %s" % (
            rep + 1, ["%.1e" % float((a - b).abs().max()) for a, b in zip(h0[0], h1[0])],
            ["%.1e" % float((a - b).abs().max()) for a, b in zip(h0[1], h1[1])],
            ["%.1e" % float((a - b).abs().max()) for a, b in zip(h0[2], h1[2])]))
    bb0 = model.backbone(img)
    for rep in range(2):
        bb1 = model.backbone(img)
        print("backbone run %d: %s" % (rep + 1, ["%.1e" % float((a - b).abs().max()) for a, b in zip(bb0, bb1)]))
    n0 = model.neck(bb0)
    for rep in range(2):
        n1 = model.neck(bb0)
        print("neck on fixed inputs run %d: %s" % (rep + 1, ["%.1e" % float((a - b).abs().max()) for a, b in zip(n0, n1)]))
