"""Development aid (GPU box): the whole detector in fp16 / bf16 at inference (`model.half()`): does every product path take half
tensors, how many detections come out against fp32, how long does a step take eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd.dota_configs import r50_model, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
dev = torch.device("cuda:0")
SIZE = int(os.environ.get("SIZE", "1024"))
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
head = model.bbox_head
with torch.no_grad():
    head.reppoints_cls_out.weight.normal_(0, 0.05); head.reppoints_cls_out.bias.fill_(-3.3)
    head.reppoints_pts_init_out.bias.copy_(torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]],
                                                        dtype=torch.float32, device=dev).reshape(-1) * 2.0)
metas = [dict(img_shape=(SIZE, SIZE, 3), pad_shape=(SIZE, SIZE, 3), scale_factor=1.0, flip=False)]
img = torch.randn(1, 3, SIZE, SIZE, device=dev)
def run(m, x, n=10):
    with torch.no_grad():
        for _ in range(3): r = m.simple_test_batch(x, metas)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = m.simple_test_batch(x, metas)
        torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) / n * 1e3
r32, t32 = run(model, img)
print("fp32: %d detections, %.2f ms per eager step" % (sum(len(c) for c in r32[0]), t32))
for dt in (torch.float16, torch.bfloat16):
    import copy
    m = copy.deepcopy(model).to(dt)
    try:
        r, t = run(m, img.to(dt))
        print("%s: %d detections, %.2f ms per eager step" % (dt, sum(len(c) for c in r[0]), t))
    except Exception as e:
        import traceback; traceback.print_exc(limit=6)
        print("%s: FAILED %r" % (dt, e))
