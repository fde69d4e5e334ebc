"""CPU stress of the device header orp_quadfast.hpp (host build, tests/host_harness): the term-queue composition the
NMS / IoU-matrix kernels run (pair classifier, per-term exact-zero screen, decision tree, per-term generic fallback)
against the oracle, bit for bit, on many seeded scenes.  usage: python tests/checks/stress_quadfast.py [million_pairs]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from orientedreppoints_amd import synthetic as S
from oracle import orp_oracle as O

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host_harness")
SO = os.path.join(HERE, "libquadfast_host.so")
subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", "-o", SO,
                       os.path.join(HERE, "quadfast_host.cpp")])
L = ctypes.CDLL(SO)
budget = float(sys.argv[1]) * 1e6 if len(sys.argv) > 1 else 20e6


def scene(seed):
    rng = np.random.RandomState(seed)
    kind = seed % 6
    n = 700
    if kind == 0:
        return S.gen_dense_scene(n, seed)[0][:, :8]
    if kind == 1:
        return S.gen_dense_scene(n, seed, clustered=False)[0][:, :8]
    if kind == 2:
        return S.gen_polys(n, seed, clustered=True)[:, :8]
    if kind == 3:
        return S.gen_polys(n, seed, clustered=True, wh=(1.0, 6.0))[:, :8]
    if kind == 4:
        d = S.gen_dense_scene(n // 2, seed)[0][:, :8].astype(np.float32)
        return np.concatenate([d, np.nextafter(d, np.float32(np.inf)), ])
    d = S.gen_polys(n, seed, clustered=True)[:, :8] - 512      # around the origin
    return d * rng.choice([1.0, 1e-3, 30.0])


done = 0; seed = 0; t0 = time.time(); tot = np.zeros(4, np.int64)
while done < budget:
    q = np.ascontiguousarray(scene(seed), np.float32)
    n = len(q)
    out = np.empty((n, n), np.float32); st = np.zeros(4, np.int64)
    guard = seed & 1
    L.host_quadterm_matrix(q.ctypes.data_as(ctypes.c_void_p), n, q.ctypes.data_as(ctypes.c_void_p), n, guard,
                           out.ctypes.data_as(ctypes.c_void_p), st.ctypes.data_as(ctypes.c_void_p))
    want = O.quad_iou_matrix(q, q, guard=bool(guard))
    bad = np.flatnonzero(out.view(np.uint32).ravel() != want.view(np.uint32).ravel())
    if len(bad):
        print("MISMATCH seed %d: %d pairs, first (%d,%d) got %r want %r" % (seed, len(bad), bad[0] // n, bad[0] % n,
              out.ravel()[bad[0]], want.ravel()[bad[0]]))
        sys.exit(1)
    done += n * n; tot += st; seed += 1
print("%d pairs over %d scenes bit-identical to the oracle in %.0f s; classifier-resolved %.1f %%, terms per unresolved "
      "pair %.2f, generic terms %.2e of evaluated" % (done, seed, time.time() - t0, 100.0 * tot[0] / done,
      tot[2] / max(done - tot[0] - tot[3], 1), tot[1] / max(tot[2], 1)))
