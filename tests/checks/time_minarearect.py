"""Development aid (GPU box): orp_minarearect_decode at one image's candidate count (5 344 point sets), HIP events around back-to-back
launches and around single launches with the device idle in between (latency, as inside an image's graph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.mmdet_ops.minarea_rect import minaerarect_decode
dev = torch.device("cuda:0")
for m in (5344, 6720, 2000, 256):
    pts = torch.from_numpy(S.gen_pointsets(m, 3).astype(np.float32)).to(dev)
    c = torch.rand(m, 2, device=dev) * 1024
    s = torch.full((m,), 8.0, device=dev)
    for _ in range(10):
        minaerarect_decode(pts, c, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        minaerarect_decode(pts, c, s)
    e1.record(); torch.cuda.synchronize()
    back = e0.elapsed_time(e1) / 50 * 1e3
    lat = []
    for _ in range(20):
        torch.cuda.synchronize()
        e0.record(); minaerarect_decode(pts, c, s); e1.record(); torch.cuda.synchronize()
        lat.append(e0.elapsed_time(e1) * 1e3)
    print("minarearect_decode %5d sets: %.1f us back to back, %.1f us median single launch (event to event)" % (m, back, float(np.median(lat))))
