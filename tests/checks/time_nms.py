"""NMS-stage timing + parity on the GPU box (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, ctypes
from orientedreppoints_amd import synthetic as S, _lib
from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
from oracle import orp_oracle as O
dev = torch.device("cuda:0")

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def prof(slot):
    tot = ctypes.c_double(0); cnt = ctypes.c_int(0)
    _lib.lib().orp_profile_read(slot, ctypes.cast(ctypes.byref(tot), ctypes.c_void_p), ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), 1)
    return tot.value / max(cnt.value, 1) * 1e3

for n, clustered in ((500, True), (2000, True), (2000, False), (5344, True), (16000, True)):
    d, _ = S.gen_dense_scene(n, 1, clustered=clustered)
    d = d.astype(np.float32)
    t = torch.from_numpy(d).to(dev)
    keep, num = rnms_device(t, 0.4)
    got = np.sort(keep[:int(num.item())].cpu().numpy())
    ok = None
    if n <= 5344:
        ok = np.array_equal(got, O.rnms(d, 0.4))
    _lib.lib().orp_profile_enable(1); prof(0); prof(1)
    us = timeit(lambda: rnms_device(t, 0.4), iters=10)
    torch.cuda.synchronize()
    m, s = prof(0), prof(1)
    _lib.lib().orp_profile_enable(0)
    print("rnms n=%d clustered=%s: total %.1f us  mask %.1f us  sweep %.1f us  kept=%d parity=%s" % (n, clustered, us, m, s, len(got), ok))

# all-pairs IoU matrix (orp_quad_iou_matrix)
for n in (2000, 5344):
    d, _ = S.gen_dense_scene(n, 1)
    t = torch.from_numpy(np.ascontiguousarray(d[:, :8], np.float32)).to(dev)
    out = torch.empty((n, n), dtype=torch.float32, device=dev)
    def run():
        rc = _lib.lib().orp_quad_iou_matrix(_lib.ptr(t), n, _lib.ptr(t), n, 8, 0, _lib.ptr(out), _lib.stream_of(t))
        assert rc == 0
    us = timeit(run, iters=10)
    print("quad_iou_matrix %d x %d: %.1f us (%.3f ns/pair)" % (n, n, us, us * 1e3 / (n * n)))
