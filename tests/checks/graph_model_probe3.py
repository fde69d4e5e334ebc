"""Development aid (GPU box): the 2nd-replay anomaly of GraphedInference in the fp16-pieces mode, variants by environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd.dota_configs import r50_model, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, GraphedInference, build_detector
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
head = model.bbox_head
with torch.no_grad():
    head.reppoints_cls_out.weight.normal_(0, 0.05)
    head.reppoints_cls_out.bias.fill_(-3.3)
B = 1
SZ = int(os.environ.get('SIZE', '256'))
metas = [dict(img_shape=(SZ, SZ, 3), pad_shape=(SZ, SZ, 3), scale_factor=1.0, flip=False)] * B
gi = GraphedInference(model, torch.randn(B, 3, SZ, SZ, device=dev), metas)
mode = os.environ.get('PROBE', 'eager_between')
for k, seed in enumerate((1, 2, 3, 4, 5)):
    img = torch.randn(B, 3, SZ, SZ, device=dev, generator=torch.Generator(device=dev).manual_seed(seed))
    got = gi(img)
    raw = gi.packed[0].cpu().numpy()[-1, :2]
    line = "replay %d: detections %s (count, overflow) %s" % (k + 1, [sum(len(c) for c in r) for r in got], raw.tolist())
    if mode == 'eager_between':
        with torch.no_grad():
            want = model.simple_test_batch(img, metas)
        same = all(np.array_equal(a, b) for gr, wr in zip(got, want) for a, b in zip(gr, wr) if a.shape == b.shape) and \
            all(a.shape == b.shape for gr, wr in zip(got, want) for a, b in zip(gr, wr))
        line += "  eager %s  identical arrays: %s" % ([sum(len(c) for c in r) for r in want], same)
    elif mode == 'alloc_between':
        junk = [torch.randn(1 << 22, device=dev) for _ in range(8)]
        torch.cuda.synchronize()
        del junk
    print(line)

# ---- do the cached weight packs survive an eager run unchanged? ----
from orientedreppoints_amd.mmdet_ops.deform_conv import _packed_weight
w = head.cls_convs[0].conv.weight
p0 = _packed_weight(w)
snap = p0.clone()
with torch.no_grad():
    model.simple_test_batch(img, metas)
torch.cuda.synchronize()
p1 = _packed_weight(w)
print("pack tensor identical object: %s, same storage: %s, bytes changed: %d of %d; tail (scale, amax) before %s after %s" % (
    p1 is p0, p1.data_ptr() == p0.data_ptr(), int((p0.view(torch.int32) != snap.view(torch.int32)).sum()), p0.numel(),
    snap[-4:-2].tolist(), p0[-4:-2].tolist()))
