"""Merge-NMS timing: fp64 polygon NMS on the GPU vs the CPU oracle port of py_cpu_nms_poly (dev aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S
from orientedreppoints_amd.dota_devkit.result_merge import py_gpu_nms_poly
from oracle import orp_oracle as O
for n in (2000, 10000, 40000):
    d = S.gen_polys(n, 5, clustered=True)
    d[:, :8] *= 4.0
    py_gpu_nms_poly(d, 0.3)                                # warm: workspace growth, first launches
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = py_gpu_nms_poly(d, 0.3)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    line = "merge NMS n=%d: GPU %.2f ms (kept %d)" % (n, (t1 - t0) * 1e3, len(k))
    if n <= 10000:
        t2 = time.perf_counter(); kc = O.py_cpu_nms_poly(d, 0.3); t3 = time.perf_counter()
        line += "  CPU port (1 core) %.1f ms  equal=%s" % ((t3 - t2) * 1e3, kc == k)
    print(line)
