"""Golden vectors for soft_rnms from THE REFERENCE'S OWN rnms_cpu.cpp (mmdet/ops/nms/src/rnms_cpu.cpp), compiled here
unmodified as a torch CPU extension (g++, -ffp-contract=off).  Run in the build container (needs /root/reference):
    python tests/golden/make_golden_softnms.py
Writes tests/golden/soft_rnms.npz: inputs + the [K,10] result for each (scene, method, threshold)."""
import os
import sys

import numpy as np
import torch
from torch.utils.cpp_extension import load

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from orientedreppoints_amd import synthetic as S  # noqa: E402

SRC = "/root/reference/mmdet/ops/nms/src/rnms_cpu.cpp"
ext = load(name="ref_rnms_cpu", sources=[SRC], extra_cflags=["-O2", "-ffp-contract=off", "-w"],
           build_directory="/tmp/ref_rnms_cpu_build" if os.makedirs("/tmp/ref_rnms_cpu_build", exist_ok=True) is None else None,
           verbose=False)
torch.set_num_threads(1)
out = {}
scenes = {
    "clustered300": S.gen_polys(300, 11, clustered=True),
    "uniform200": S.gen_polys(200, 12),
    "dense_offset400": S.gen_dense_scene(400, 13)[0],
}
cases = []
for name, d in scenes.items():
    d = np.ascontiguousarray(d, np.float32)
    out["in_" + name] = d
    for method, thr, sigma, min_score in ((0, 0.4, 0.5, 1e-3), (1, 0.3, 0.5, 1e-3), (2, 0.3, 0.5, 0.05), (1, 0.1, 0.5, 0.2)):
        res = ext.soft_rnms(torch.from_numpy(d.copy()), thr, method, sigma, min_score).numpy()
        key = "%s_m%d_t%g_s%g_ms%g" % (name, method, thr, sigma, min_score)
        out["out_" + key] = res
        cases.append(key)
out["cases"] = np.array(cases)
np.savez_compressed(os.path.join(HERE, "soft_rnms.npz"), **out)
print("wrote", len(cases), "cases;", {k: out["out_" + k].shape for k in cases[:4]})
