"""Golden values for the Task1 evaluation from THE REFERENCE'S OWN DOTA_devkit/dota_evaluation_task1.py (voc_eval) run
here on the CPU: detections = the merged golden files of make_golden_merge.py, ground truth = synthetic labelTxt files
written here (the objects the detections were synthesised from, jittered; every 7th one flagged difficult).
The reference module needs `np.bool` (removed from numpy) and matplotlib (absent): both are shimmed before import, the
reference's `polyiou` SWIG module is built from its own sources into /tmp as in make_golden_merge.py.
    python tests/golden/make_golden_eval.py      -> tests/golden/eval/{labelTxt/*.txt, imageset.txt, voc_eval.npz}
"""
import contextlib
import io
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
if not hasattr(np, "bool"):
    np.bool = bool                                         # the reference predates numpy 1.24
mpl = types.ModuleType("matplotlib"); plt = types.ModuleType("matplotlib.pyplot")
mpl.pyplot = plt; sys.modules["matplotlib"] = mpl; sys.modules["matplotlib.pyplot"] = plt
import dota_evaluation_task1 as REFEVAL  # noqa: E402  (the reference module)

sys.path.insert(0, os.path.join(HERE, "..", ".."))
from orientedreppoints_amd import synthetic as S  # noqa: E402

out = os.path.join(HERE, "eval")
shutil.rmtree(out, ignore_errors=True)
os.makedirs(os.path.join(out, "labelTxt"))
rng = np.random.RandomState(5)
classes = ["plane", "ship", "small-vehicle"]
images = ("P0001", "P0002", "P0003")                      # P0003: no detections, ground truth only
lines = {img: [] for img in images}
for ci, cls in enumerate(classes):
    for img in images:
        objs = S.gen_polys(160, 100 * ci + (img == "P0002") + 7 * (img == "P0003"), clustered=True, wh=(12.0, 90.0))
        objs[:, :8] *= 2.4
        for k, o in enumerate(objs[::2]):                  # half of the objects are annotated
            q = o[:8] + rng.normal(0, 1.5, 8)
            lines[img].append("%s %s %d" % (" ".join("%.1f" % v for v in q), cls, 1 if k % 7 == 0 else 0))
for img in images:
    with open(os.path.join(out, "labelTxt", img + ".txt"), "w") as f:
        f.write("imagesource:synthetic\ngsd:null\n" + "\n".join(lines[img]) + "\n")
with open(os.path.join(out, "imageset.txt"), "w") as f:
    f.write("\n".join(images) + "\n")

detpath = os.path.join(HERE, "merge", "merged", "Task1_{:s}.txt")
annopath = os.path.join(out, "labelTxt", "{:s}.txt")
res = {}
for cls in classes:
    for thr in (0.5, 0.3):
        for m07 in (False, True):
            with contextlib.redirect_stdout(io.StringIO()):
                rec, prec, ap = REFEVAL.voc_eval(detpath, annopath, os.path.join(out, "imageset.txt"), cls,
                                                 ovthresh=thr, use_07_metric=m07)
            key = "%s_%02d_%d" % (cls, int(thr * 10), int(m07))
            res["rec_" + key] = rec; res["prec_" + key] = prec; res["ap_" + key] = np.float64(ap)
            print(key, "nd", len(rec), "ap %.6f" % ap, "final recall %.4f" % (rec[-1] if len(rec) else 0))
np.savez_compressed(os.path.join(out, "voc_eval.npz"), **res)
