"""Golden files for the merge workflow from THE REFERENCE'S OWN DOTA_devkit/ResultMerge.py (mergebypoly) run here on
the CPU: synthetic per-class Task1 patch result files -> merged files.  `dota_utils` imports shapely (absent): a stub
`shapely.geometry` is registered before import -- ResultMerge only uses dota_utils' two path helpers.  The reference's
`polyiou` SWIG module is built from its own sources (polyiou.cpp + polyiou_wrap.cxx) into /tmp.
    python tests/golden/make_golden_merge.py        -> tests/golden/merge/{raw,merged}/Task1_*.txt
"""
import os
import shutil
import subprocess
import sys
import sysconfig
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/DOTA_devkit"
BUILD = "/tmp/ref_polyiou_build"
os.makedirs(BUILD, exist_ok=True)
so = os.path.join(BUILD, "_polyiou" + sysconfig.get_config_var("EXT_SUFFIX"))
if not os.path.exists(so):
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-I" + sysconfig.get_paths()["include"],
                           os.path.join(REF, "polyiou.cpp"), os.path.join(REF, "polyiou_wrap.cxx"), "-o", so])
    shutil.copy(os.path.join(REF, "polyiou.py"), BUILD)
sys.path.insert(0, BUILD)
sys.path.insert(0, REF)
shp = types.ModuleType("shapely"); geo = types.ModuleType("shapely.geometry")
shp.geometry = geo; sys.modules["shapely"] = shp; sys.modules["shapely.geometry"] = geo
import ResultMerge  # noqa: E402  (the reference module)

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from orientedreppoints_amd import synthetic as S  # noqa: E402

raw, merged = os.path.join(HERE, "merge", "raw"), os.path.join(HERE, "merge", "merged")
for d in (raw, merged):
    shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
rng = np.random.RandomState(0)
# two original images cut into overlapping 1024 patches with stride 824 at rates 1 and 0.5; objects near patch borders
# are detected in several patches with slightly different coordinates -> the merge NMS has real work
for ci, cls in enumerate(["plane", "ship", "small-vehicle"]):
    lines = []
    for img in ("P0001", "P0002"):
        objs = S.gen_polys(160, 100 * ci + (img == "P0002"), clustered=True, wh=(12.0, 90.0))
        objs[:, :8] *= 2.4                                   # spread over a ~2460^2 image
        for rate in ("1", "0.5"):
            r = float(rate)
            for px in range(0, 2048, 824):
                for py in range(0, 2048, 824):
                    for o in objs:
                        p = o[:8] * r
                        cx, cy = p[0::2].mean(), p[1::2].mean()
                        if px <= cx < px + 1024 and py <= cy < py + 1024:
                            q = p - np.tile([px, py], 4) + rng.normal(0, 0.8, 8)
                            sc = np.clip(o[8] + rng.normal(0, 0.05), 0.01, 1.0)
                            lines.append("%s__%s__%d___%d %.3f %s" % (img, rate, px, py, sc, " ".join("%.1f" % v for v in q)))
    with open(os.path.join(raw, "Task1_%s.txt" % cls), "w") as f:
        f.write("\n".join(lines) + "\n")
ResultMerge.mergebypoly(raw, merged)
print({f: sum(1 for _ in open(os.path.join(raw, f))) for f in sorted(os.listdir(raw))},
      {f: sum(1 for _ in open(os.path.join(merged, f))) for f in sorted(os.listdir(merged))})
