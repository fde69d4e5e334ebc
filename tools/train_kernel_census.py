"""Development aid (GPU box): kernels launched per phase of one training step (torch.profiler device events grouped by
record_function ranges around backbone / neck / head forward, the loss and its sub-phases, backward, optimizer)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity, record_function
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.dota_configs import r50_model, train_cfg, test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector
import orientedreppoints_amd.mmdet_models.orientedreppoints_head_train as HT
dev = torch.device("cuda:0")
B, K = 2, 64
torch.manual_seed(0)
model = build_detector(ConfigDict(r50_model), train_cfg=ConfigDict(train_cfg), test_cfg=ConfigDict(test_cfg)).to(dev).train()
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=1e-4, momentum=0.9, weight_decay=1e-4)
img = torch.randn(B, 3, 1024, 1024, device=dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)] * B
gts = [torch.from_numpy(S.gen_polys(K, 40 + i, wh=(16, 120))[:, :8].astype(np.float32)).to(dev) for i in range(B)]
labels = [torch.randint(1, 16, (K,), device=dev) for _ in range(B)]


def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function(tag):
            return f(*a, **k)
    setattr(obj, name, g)


wrap(model.backbone, 'forward', 'P:backbone')
if getattr(model, 'neck', None) is not None:
    wrap(model.neck, 'forward', 'P:neck')
wrap(model.bbox_head, 'forward', 'P:head_forward')
for fn in ('init_pointset_target', 'refine_pointset_target', 'points_quality_assessment', 'point_samples_selection',
           'offset_to_pts', 'get_points', 'init_loss_single'):
    if hasattr(HT, fn):
        wrap(HT, fn, 'P:loss.' + fn)
wrap(HT, 'head_loss', 'P:loss')


def step():
    losses = model(img, metas, return_loss=True, gt_bboxes=gts, gt_labels=labels)
    total = sum(sum(v) if isinstance(v, (list, tuple)) else v for v in losses.values())
    opt.zero_grad(set_to_none=True)
    with record_function('P:backward'):
        total.backward()
    with record_function('P:optimizer'):
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
ranges = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith('P:')]
# launches are counted on the CPU side (hip*Launch* / hipMemcpyAsync / hipMemsetAsync runtime events) inside the innermost
# P: range; the op that issued them is the closest enclosing aten:: / autograd event
count = collections.Counter(); ops = collections.defaultdict(collections.Counter)
cpu_ops = [(e.name, e.time_range.start, e.time_range.end) for e in ev
           if e.device_type.name == 'CPU' and (e.name.startswith('aten::') or 'Backward' in e.name or e.name.startswith('autograd::'))]
cpu_ops.sort(key=lambda t: t[1])
import bisect
starts = [t[1] for t in cpu_ops]
for e in ev:
    if e.device_type.name != 'CPU':
        continue
    n = e.name
    if not (('Launch' in n and n.startswith('hip')) or n in ('hipMemcpyAsync', 'hipMemsetAsync', 'hipMemcpyWithStream')):
        continue
    t = e.time_range.start
    inner = None
    for nm, a, b in ranges:
        if a <= t <= b and (inner is None or (b - a) < (inner[2] - inner[1])):
            inner = (nm, a, b)
    tag = inner[0] if inner else 'P:other'
    count[tag] += 1
    # innermost enclosing op
    i = bisect.bisect_right(starts, t) - 1
    best = None
    while i >= 0 and i > bisect.bisect_right(starts, t) - 400:
        nm, a, b = cpu_ops[i]
        if a <= t <= b:
            best = nm; break
        i -= 1
    ops[tag][best or n] += 1
tot = sum(count.values())
print("launches in one step: %d" % tot)
for tag, n in count.most_common():
    print("%-40s %5d launches" % (tag, n))
    for nm, c in ops[tag].most_common(14):
        print("        %4d  %s" % (c, nm))
