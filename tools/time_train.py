"""One training step of the detector at BASELINE configs[2] shapes (R-50 FPN, 1024x1024, 2 img/GPU, APAA on) -- timing
aid: forward(loss) / backward wall time with a per-phase split."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.dota_configs import r50_model, train_cfg, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).to(dev)
model.train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=1e-4)
img = torch.randn(B, 3, 1024, 1024, device=dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)] * B
gts = [torch.from_numpy(S.gen_polys(K, 40 + i, wh=(16, 120))[:, :8].astype(np.float32)).to(dev) for i in range(B)]
labels = [torch.randint(1, 16, (K,), device=dev) for _ in range(B)]
def step():
    t0 = time.perf_counter()
    losses = model(img, metas, return_loss=True, gt_bboxes=gts, gt_labels=labels)
    total = sum(sum(v) if isinstance(v, (list, tuple)) else v for v in losses.values())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    total.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return t1 - t0, t2 - t1, t3 - t2, float(total)
for _ in range(2): step()
acc = np.zeros(3)
for _ in range(iters):
    f, b, o, tot = step(); acc += (f, b, o)
acc /= iters
print("train step B=%d K=%d: forward+loss %.1f ms  backward %.1f ms  optimizer %.1f ms  total %.1f ms  (%.2f img/s)  loss %.3f"
      % (B, K, acc[0] * 1e3, acc[1] * 1e3, acc[2] * 1e3, acc.sum() * 1e3, B / acc.sum(), tot))
