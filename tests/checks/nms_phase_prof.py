"""Development aid (GPU box): per-phase shader-clock cycles of the rotated-NMS mask kernel, from a library variant built
with -DORP_NMS_PHASE_PROF (python tools/build_variant.py phaseprof orp_nms.hip -DORP_NMS_PHASE_PROF; run with
ORP_HIP_LIB=build_variants/liborp_hip_phaseprof.so).  Cycles are those of thread 0 of each workgroup, summed over tiles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from orientedreppoints_amd import synthetic as S, _lib
from orientedreppoints_amd.mmdet_ops.nms_wrapper import rnms_device
dev = torch.device("cuda:0")
L = _lib.lib()
rd = L.orp_nms_phase_prof_read
rd.restype = ctypes.c_int
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = {0: "stage", 1: "phaseA", 2: "drain", 3: "write", 8: "B1", 9: "B2", 10: "B3"}
for ncls in (15, 1):
    for n in (2000,):
        d, _ = S.gen_dense_scene(n, 1, num_classes=ncls, clustered=True)
        t = torch.from_numpy(d.astype(np.float32)).to(dev)
        for _ in range(3):
            rnms_device(t, 0.4)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        rd(buf, 1)
        iters = 10
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            rnms_device(t, 0.4)
        e1.record(); torch.cuda.synchronize()
        rd(buf, 1)
        v = [x / iters for x in buf]
        tiles, wgs, chunks = v[4], v[6], max(v[11], 1)
        print("classes=%d n=%d: %.1f us per call; tiles %d workgroups %d; cycles per tile: %s; per workgroup whole kernel %.0f; "
              "chunks/tile %.2f B2 iterations/tile %.2f; per chunk: B1 %.0f B2 %.0f B3 %.0f"
              % (ncls, n, e0.elapsed_time(e1) / iters * 1e3, tiles, wgs,
                 " ".join("%s %.0f" % (names[k], v[k] / tiles) for k in (0, 1, 2, 3)), v[5] / max(wgs, 1),
                 chunks / tiles, v[12] / tiles, v[8] / chunks, v[9] / chunks, v[10] / chunks))

# ---- timeline of ONE launch: start / end (100 MHz wall clock) of every workgroup -> resident workgroups over time -----
raw = L.orp_nms_phase_prof_raw
raw.restype = ctypes.c_int
raw.argtypes = [ctypes.c_void_p]
for ncls in (15,):
    d, _ = S.gen_dense_scene(2000, 1, num_classes=ncls, clustered=True)
    t = torch.from_numpy(d.astype(np.float32)).to(dev)
    rd((ctypes.c_ulonglong * 16)(), 1)
    rnms_device(t, 0.4)
    torch.cuda.synchronize()
    buf = np.zeros((4096, 16), np.uint64)
    raw(buf.ctypes.data_as(ctypes.c_void_p))
    st, en, cyc = buf[:, 13].astype(np.int64), buf[:, 14].astype(np.int64), buf[:, 5].astype(np.int64)
    ok = en > 0
    st, en, cyc = st[ok], en[ok], cyc[ok]
    t0 = st.min()
    st, en = (st - t0) / 100.0, (en - t0) / 100.0          # us
    print("timeline (classes=%d): %d workgroups; first start 0, last start %.1f us, last end %.1f us; mean duration %.1f us (p10 %.1f p50 %.1f p90 %.1f max %.1f); effective clock %.2f GHz"
          % (ncls, len(st), st.max(), en.max(), (en - st).mean(), *np.percentile(en - st, [10, 50, 90]), (en - st).max(),
             (cyc / np.maximum(en - st, 1e-9)).mean() / 1e3))
    edges = np.arange(0, en.max() + 5, 5.0)
    occ = [int(((st < b + 5) & (en > b)).sum()) for b in edges]
    print("  resident workgroups per 5 us bin:", occ)
