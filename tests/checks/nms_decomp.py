"""Development aid (GPU box): time decomposition of the rotated-NMS stage via the ORP_NMS_DBG / ORP_NMS_ROWS switches.
Each configuration runs in a fresh process (the switches are read once)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
code = r'''
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath("%s")))))
import numpy as np, torch, ctypes
from orientedreppoints_amd import synthetic as S, _lib
from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
dev = torch.device("cuda:0")
def prof(slot):
    tot = ctypes.c_double(0); cnt = ctypes.c_int(0)
    _lib.lib().orp_profile_read(slot, ctypes.cast(ctypes.byref(tot), ctypes.c_void_p), ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), 1)
    return tot.value / max(cnt.value, 1) * 1e3
ncls = int(os.environ.get('ORP_DECOMP_CLASSES', '15'))
for n in (2000, 5344):
    d, _ = S.gen_dense_scene(n, 1, num_classes=ncls, clustered=True)
    t = torch.from_numpy(d.astype(np.float32)).to(dev)
    for _ in range(3): rnms_device(t, 0.4)
    torch.cuda.synchronize()
    _lib.lib().orp_profile_enable(1); prof(0); prof(1); prof(2)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): rnms_device(t, 0.4)
    e1.record(); torch.cuda.synchronize()
    print("  n=%%d total %%.1f us sort+prep %%.1f mask %%.1f sweep %%.1f" %% (n, e0.elapsed_time(e1) / 20 * 1e3, prof(2), prof(0), prof(1)))
    _lib.lib().orp_profile_enable(0)
''' % os.path.abspath(__file__)
cfgs = [{}] + [{'ORP_NMS_DBG': v} for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else [{}]
rows = sys.argv[2].split(',') if len(sys.argv) > 2 else []
for env in cfgs + [{'ORP_NMS_ROWS': r} for r in rows]:
    print(env, flush=True)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, '-c', code], env=e)
