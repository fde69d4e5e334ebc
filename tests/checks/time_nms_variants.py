"""Development aid (GPU box): the single-image rotated NMS (2 000 class-offset boxes, dense scene) -- stage time and keep-set parity
with the oracle -- for a library variant (ORP_HIP_LIB); ORP_NMS_ROWS=4 keeps the tile at 16 rows (needed by -DORP_TILE_ROWS=16 builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
from oracle import orp_oracle as O
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("ORP_HIP_LIB", "in-tree"))
for n, clustered in ((2000, True), (2000, False), (1000, True)):
    d, _ = S.gen_dense_scene(n, 1, clustered=clustered)
    d = d.astype(np.float32)
    t = torch.from_numpy(d).to(dev)
    keep, num = rnms_device(t, 0.4)
    got = np.sort(keep[:int(num.item())].cpu().numpy())
    ok = np.array_equal(got, np.sort(O.rnms(d, 0.4)))
    for _ in range(10):
        rnms_device(t, 0.4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        rnms_device(t, 0.4)
    e1.record(); torch.cuda.synchronize()
    print("[%s] n=%d clustered=%s: %.1f us per rnms call (3 kernels, back to back), keep set == oracle: %s" % (tag, n, clustered, e0.elapsed_time(e1) / 50 * 1e3, ok))
