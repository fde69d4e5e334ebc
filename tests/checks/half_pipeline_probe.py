"""Dev probe (GPU): a model.half() detector from ONE captured graph, then `depth` graphs in flight.  DET=1 restricts the library to its
reproducible solvers (torch.backends.cudnn.deterministic): with that AND depth 4 the device stopped making progress -- the combination
PipelinedInference refuses since round 6 (this script bypasses the guard); without it the model runs four deep at ~338 images/s.
   [DET=1] [ORP_DCNH_WS=0|1] python tests/checks/half_pipeline_probe.py [depth] [float16|bfloat16]"""
import copy, faulthandler, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
faulthandler.dump_traceback_later(70, repeat=False, file=sys.stderr)
import numpy as np
import torch
import bench
from orientedreppoints_amd.mmdet_models import ConfigDict, GraphedInference, PipelinedInference, build_detector

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
half = getattr(torch, sys.argv[2]) if len(sys.argv) > 2 else torch.float16
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = build_detector(ConfigDict(bench.MODELS["r50"]), train_cfg=None, test_cfg=ConfigDict(copy.deepcopy(bench.TEST_CFG))).to(dev).eval()
img = torch.randn(1, 3, 1024, 1024, generator=torch.Generator(device="cpu").manual_seed(4321)).to(dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)]
bench.calibrate_head(model, img)
model, img = model.to(half), img.to(half)
if os.environ.get('DET', '0') == '1':
    torch.backends.cudnn.deterministic = True           # the library's reproducible solvers only (what bench.quick_config switches to)
same = lambda ra, rb: all(a.shape == b.shape and np.array_equal(a, b) for r, q in zip(ra, rb) for a, b in zip(r, q))   # noqa: E731
with torch.no_grad():
    ref = model.simple_test_batch(img, metas)
gi = GraphedInference(model, img, metas)
for _ in range(3):
    r = gi(img)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gi(img)
torch.cuda.synchronize()
print("%s ORP_DCNH_WS=%s: one graph %.0f images/s, %d detections, identical to eager %s" % (
    half, os.environ.get("ORP_DCNH_WS", "default"), 10 / (time.perf_counter() - t0), sum(len(c) for c in ref[0]), same(r, ref)), flush=True)
del gi
if depth > 1:
    pi = PipelinedInference(model, img, metas, depth=depth, _allow_half=True)
    got = [r for r in (pi.submit(img) for _ in range(3 + depth)) if r is not None] + pi.flush()
    print("  warm-up with %d graphs in flight done, identical %s" % (depth, all(same(g, ref) for g in got)), flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = sum(pi.submit(img) is not None for _ in range(10)) + len(pi.flush())
    torch.cuda.synchronize()
    print("  %d graphs in flight: %.0f images/s" % (depth, n / (time.perf_counter() - t0)), flush=True)
