import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from orientedreppoints_amd.dota_configs import r50_model, test_cfg as TEST_CFG
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(TEST_CFG)).to(dev).eval()
g = torch.Generator(device='cpu').manual_seed(1234)
img = torch.randn(1, 3, 1024, 1024, generator=g).to(dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)]
bench.calibrate_head(model, img)
with torch.no_grad():
    a = model.simple_test_batch(img, metas)[0]
    model.test_cfg['static_postprocess'] = False
    b = model.simple_test_batch(img, metas)[0]
    b2 = model.simple_test_batch(img, metas)[0]
print('static', sum(len(x) for x in a), 'dynamic', sum(len(x) for x in b), 'dynamic again', sum(len(x) for x in b2))
for i, (x, y) in enumerate(zip(a, b)):
    if x.shape != y.shape or not np.array_equal(x, y):
        print('class', i, x.shape, y.shape, 'max abs diff', np.abs(x - y).max() if x.shape == y.shape else None)
