import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from orientedreppoints_amd.dota_configs import r50_model, test_cfg as TEST_CFG
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(TEST_CFG)).to(dev).eval()
img = torch.randn(1, 3, 1024, 1024, device=dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)]
bench.calibrate_head(model, img)
head = model.bbox_head
with torch.no_grad():
    feats_b = model.backbone(img)
    feats = model.neck(feats_b)
    outs = head(feats)
def try_capture(name, fn):
    try:
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2): fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            fn()
        g.replay(); torch.cuda.synchronize()
        print(name, 'OK')
    except Exception as e:
        print(name, 'FAILED:', str(e).split('\n')[0][:120])
        torch.cuda.synchronize()
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi, minaerarect
from orientedreppoints_amd.mmdet_ops.fused_norm import group_norm_act_multi, bn_act
from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_batched_device
import numpy as np
from orientedreppoints_amd import synthetic as S
try_capture('conv only', lambda: model.backbone.conv1(img))
try_capture('bn_act', lambda: bn_act(model.backbone.conv1(img).contiguous(), model.backbone.bn1))
try_capture('backbone', lambda: model.backbone(img))
try_capture('neck', lambda: model.neck(feats_b))
try_capture('gn multi', lambda: group_norm_act_multi([f.clone() for f in feats], head.cls_convs[0].norm))
w = head.reppoints_cls_conv.weight
offs = [torch.zeros(1, 18, f.size(2), f.size(3), device=dev) for f in feats]
try_capture('dcn multi', lambda: deform_conv_forward_multi(list(feats), offs, w, 1, 1, 1))
try_capture('head', lambda: head(feats))
d = torch.from_numpy(S.gen_dense_scene(2000, 1)[0].astype(np.float32)).to(dev)
seg = torch.tensor([0, 2000], dtype=torch.int32, device=dev)
try_capture('rnms_batched', lambda: rnms_batched_device(d, seg, 4096, 0.4))
pts = torch.from_numpy(S.gen_pointsets(512, 0).astype(np.float32)).to(dev)
try_capture('minaerarect', lambda: minaerarect(pts))
try_capture('get_bboxes static', lambda: head.get_bboxes(*(tuple(outs) + (metas, model.test_cfg, False)), static=True))
