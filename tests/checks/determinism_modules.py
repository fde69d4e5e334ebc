"""Development aid (GPU box): WHICH module of the eager inference step is not bitwise reproducible run to run?  Forward hooks on
every leaf module record a checksum of inputs and outputs; a module whose inputs agree between two runs and whose output
does not is a source (SIZE=1536 python tests/checks/determinism_modules.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from orientedreppoints_amd.dota_configs import test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, build_detector

size = int(os.environ.get('SIZE', '1536'))
if os.environ.get('DETERMINISTIC', '0') == '1':
    torch.backends.cudnn.deterministic = True          # MIOpen: only solvers that are run-to-run reproducible
dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_detector(ConfigDict(Bn.MODELS[os.environ.get('MODEL', 'r50')]), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
img = torch.randn(1, 3, size, size, device=dev)
log = []


def csum(t):
    if isinstance(t, torch.Tensor):
        return [int(t.detach().contiguous().view(torch.int32).to(torch.int64).sum())] if t.dtype == torch.float32 else []
    if isinstance(t, (list, tuple)):
        return [v for x in t for v in csum(x)]
    return []


def hook(name):
    def f(mod, inp, out):
        log.append((name, type(mod).__name__, tuple(csum(inp)), tuple(csum(out)),
                    tuple(inp[0].shape) if inp and isinstance(inp[0], torch.Tensor) else None))
    return f


for n, m in model.named_modules():
    if not list(m.children()):
        m.register_forward_hook(hook(n))
runs = []
for rep in range(int(os.environ.get('RUNS', '6'))):
    log.clear()
    with torch.no_grad():
        c = model.backbone(img); f = model.neck(c); model.bbox_head(f)
    torch.cuda.synchronize()
    runs.append(list(log))
bad = {}
for r in runs[1:]:
    for a, b in zip(runs[0], r):
        assert a[0] == b[0]
        if a[2] == b[2] and a[3] != b[3]:
            bad[(a[0], a[1], a[4])] = bad.get((a[0], a[1], a[4]), 0) + 1
print("size %d: %d hooked calls per run; modules with identical inputs and differing outputs over %d re-runs:" % (size, len(runs[0]), len(runs) - 1))
for k, v in bad.items():
    m = dict(model.named_modules())[k[0]]
    print("  %s (%s) input %s  %s  -- %d of %d" % (k[0], k[1], k[2], m, v, len(runs) - 1))
if not bad:
    print("  none")

import time
with torch.no_grad():
    for _ in range(3):
        model.backbone(img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.backbone(img)
    torch.cuda.synchronize()
print("backbone at %d: %.3f ms (cudnn.deterministic=%s)" % (size, (time.perf_counter() - t0) * 100, torch.backends.cudnn.deterministic))
