"""Development aid (GPU box): one training step of the detector under torch.autocast(fp16 / bf16)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.dota_configs import r50_model, train_cfg, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
dev = torch.device("cuda:0")
dt = dict(fp16=torch.float16, bf16=torch.bfloat16)[sys.argv[1] if len(sys.argv) > 1 else "fp16"]
B, K = 2, 64
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).to(dev).train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=1e-4)
scaler = torch.amp.GradScaler("cuda", enabled=(dt == torch.float16))
img = torch.randn(B, 3, 1024, 1024, device=dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)] * B
gts = [torch.from_numpy(S.gen_polys(K, 40 + i, wh=(16, 120))[:, :8].astype(np.float32)).to(dev) for i in range(B)]
labels = [torch.randint(1, 16, (K,), device=dev) for _ in range(B)]
def step():
    with torch.autocast(device_type="cuda", dtype=dt):
        losses = model(img, metas, return_loss=True, gt_bboxes=gts, gt_labels=labels)
    total = sum(sum(v) if isinstance(v, (list, tuple)) else v for v in losses.values())
    opt.zero_grad(set_to_none=True)
    scaler.scale(total.float().sum()).backward()
    scaler.step(opt); scaler.update()
    return float(total.float().sum())
for i in range(3):
    l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(5):
    l = step()
torch.cuda.synchronize()
print("autocast %s: %.1f ms per step, loss %.3f" % (dt, (time.perf_counter() - t0) / 5 * 1e3, l))
