"""Development aid (GPU box): the bench's throughput mode under soak -- the SAME calibrated 1024^2 scene submitted N times
through PipelinedInference (4 graphs in flight); every result must equal the eager result exactly."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench as Bn
from orientedreppoints_amd.dota_configs import test_cfg
from orientedreppoints_amd.mmdet_models import ConfigDict, PipelinedInference, GraphedInference, build_detector

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = build_detector(ConfigDict(Bn.MODELS['r50']), train_cfg=None, test_cfg=ConfigDict(test_cfg)).to(dev).eval()
img = torch.randn(1, 3, 1024, 1024, device=dev)
metas = [dict(img_shape=(1024, 1024, 3), pad_shape=(1024, 1024, 3), scale_factor=1.0, flip=False)]
Bn.calibrate_head(model, img)
with torch.no_grad():
    want = model.simple_test_batch(img, metas)
    again = model.simple_test_batch(img, metas)
nd = sum(len(c) for r in want for c in r)
print("eager dets", nd, "eager repeat identical:", all(np.array_equal(a, b) for r, s in zip(want, again) for a, b in zip(r, s)))
N = int(os.environ.get('SOAK_N', '300'))
for name, runner in (('graphed', lambda: GraphedInference(model, img, metas)), ('pipelined', lambda: PipelinedInference(model, img, metas, depth=4))):
    r = runner()
    if name == 'graphed':
        got = [r(img) for _ in range(N)]
    else:
        got = [x for x in (r.submit(img) for _ in range(N)) if x is not None] + r.flush()
    bad_count = sum(1 for g in got if sum(len(c) for rr in g for c in rr) != nd)
    bad_vals = sum(1 for g in got if not all(a.shape == b.shape and np.array_equal(a, b) for rr, ww in zip(g, want) for a, b in zip(rr, ww)))
    first = next((i for i, g in enumerate(got) if sum(len(c) for rr in g for c in rr) != nd), None)
    print("%s: %d results, wrong detection count %d (first at %s), not bit-identical %d  [KSPLIT env %s, lib %s]"
          % (name, len(got), bad_count, first, bad_vals, os.environ.get('ORP_DCN_KSPLIT', 'default'), os.environ.get('ORP_HIP_LIB', 'in-tree')))
    del r
