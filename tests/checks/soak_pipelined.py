import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from orientedreppoints_amd.dota_configs import r50_model, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, PipelinedInference, build_detector
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
head = model.bbox_head
with torch.no_grad():
    head.reppoints_cls_out.weight.normal_(0, 0.05)
    head.reppoints_cls_out.bias.fill_(-3.3)
    head.reppoints_pts_init_out.bias.copy_(torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 0], [0, 1], [1, -1], [1, 0], [1, 1]], dtype=torch.float32, device=dev).reshape(-1) * 2.0)
metas = [dict(img_shape=(512, 512, 3), pad_shape=(512, 512, 3), scale_factor=1.0, flip=False)]
imgs = [torch.randn(1, 3, 512, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) for i in range(5)]
with torch.no_grad():
    want = [model.simple_test_batch(im, metas) for im in imgs]
pi = PipelinedInference(model, imgs[0], metas, depth=4)
bad = 0; n = 0; got = []
order = np.random.RandomState(0).randint(0, 5, size=1500)
for j in order:
    r = pi.submit(imgs[j])
    if r is not None: got.append(r)
got += pi.flush()
assert len(got) == len(order)
for j, g in zip(order, got):
    for gr, wr in zip(g, want[j]):
        for a, b in zip(gr, wr):
            if a.shape != b.shape or not np.allclose(a, b, rtol=1e-4, atol=1e-2): bad += 1
print("soak: %d submits, mismatching class arrays: %d, dets per image %s" % (len(order), bad, [sum(len(c) for r in w for c in r) for w in want]))
