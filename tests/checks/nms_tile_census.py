"""Development aid (CPU): how the rotated-NMS mask kernel's work is distributed over its tiles.  Uses the host build of
orp_quadfast.hpp (tests/host_harness): per pair, does the classifier resolve it (phase A) and how many fan terms does the
per-term screen leave (phase B2 work).  Prints totals and the per-tile distribution for a tile shape.
usage: python tests/checks/nms_tile_census.py [n] [classes] [rows_per_tile]"""
import ctypes, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from orientedreppoints_amd import synthetic as S

SRC = os.path.join(ROOT, "tests", "host_harness", "quadfast_host.cpp")
SO = os.path.join(ROOT, "tests", "host_harness", "libquadfast_host.so")
subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
L = ctypes.CDLL(SO)


def census(d):
    order = np.lexsort((np.arange(len(d)), -d[:, 8]))
    a = np.ascontiguousarray(d[order, :8], np.float32)
    n = len(a)
    out = np.zeros((n, n), np.uint8)
    L.host_pair_census(a.ctypes.data_as(ctypes.c_void_p), n, out.ctypes.data_as(ctypes.c_void_p))
    return out


def report(name, d, rows):
    c = census(d)
    n = len(d)
    iu = np.triu_indices(n, 1)
    v = c[iu]
    pend = v > 0
    terms = np.where(pend, v.astype(np.int64) - 1, 0)
    print("%s: n=%d pairs=%d unresolved=%d (%.1f%%) screened-empty=%d terms=%d (%.2f per unresolved pair)"
          % (name, n, len(v), pend.sum(), 100.0 * pend.mean(), (v == 1).sum(), terms.sum(), terms.sum() / max(1, pend.sum())))
    # per tile (rows x 64 columns), upper-triangular tiles only
    cb = (n + 63) // 64
    T = np.where(np.triu(np.ones((n, n), bool), 1), np.where(c > 0, c.astype(np.int64) - 1, 0), 0)
    P = np.triu(c > 0, 1)
    tt, tp = [], []
    for r0 in range(0, n, rows):
        for cc in range(r0 // 64, cb):
            tt.append(T[r0:r0 + rows, cc * 64:(cc + 1) * 64].sum())
            tp.append(P[r0:r0 + rows, cc * 64:(cc + 1) * 64].sum())
    tt, tp = np.array(tt), np.array(tp)
    q = lambda a, p: int(np.percentile(a, p))
    print("  tiles %dx64: %d; pending pairs/tile mean %.0f p50 %d p90 %d p99 %d max %d; terms/tile mean %.0f p50 %d p90 %d p99 %d max %d"
          % (rows, len(tt), tp.mean(), q(tp, 50), q(tp, 90), q(tp, 99), tp.max(), tt.mean(), q(tt, 50), q(tt, 90), q(tt, 99), tt.max()))
    # B2 iterations per tile (256 lanes per iteration, chunks of 256 pairs) and an ideal-balance bound
    it = np.ceil(tt / 256.0)
    print("  B2 iterations: total %d, max per tile %d; tiles with zero pending pairs: %d" % (it.sum(), it.max(), (tp == 0).sum()))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    classes = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    for clustered in (True, False):
        d, _ = S.gen_dense_scene(n, 1, num_classes=classes, clustered=clustered)
        report("classes=%d clustered=%s" % (classes, clustered), d.astype(np.float32), rows)
