"""Bitwise comparison of orp_minarearect with the oracle on ~1 M point sets (GPU box).

Round-5 verdict, weak 1 / next 2a: rounds 1-5 held minaerarect to a relative 1e-4 and exempted "proven ties" in the APAA quality
test.  This script counts the sets whose 8 output floats differ in ANY bit, family by family:
  random      -- synthetic.gen_pointsets (jittered rotated grids, what a trained head emits)
  grid_axis   -- exact regular 3x3 grids, axis aligned (what the initial stage emits: dcn_base_offset * scale)
  grid_rot    -- exact regular grids rotated by a random angle and rounded to float
  grid_int    -- integer-coordinate lattices (exact ties by construction)
  collinear / duplicate / tiny -- degenerate hulls
and, for the first mismatches, prints the point set and both results.  It also checks the device build of orp_libm.hpp against
the host C library on every float of (-4, 4).
    python tests/checks/minarearect_bits.py [n_per_family]
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from orientedreppoints_amd import _lib, synthetic as S                       # noqa: E402
from orientedreppoints_amd.mmdet_ops import minaerarect                      # noqa: E402
from oracle import orp_oracle as O                                           # noqa: E402


def families(n, seed=0):
    rng = np.random.RandomState(seed)
    g = np.array([[x, y] for y in (-1.0, 0.0, 1.0) for x in (-1.0, 0.0, 1.0)])
    fam = {}
    fam["random"] = S.gen_pointsets(n, seed + 1)
    sx, sy = rng.uniform(0.5, 40, (n, 1)), rng.uniform(0.5, 40, (n, 1))
    c = rng.uniform(0, 1024, (n, 1, 2))
    ax = np.stack([g[None, :, 0] * sx, g[None, :, 1] * sy], 2) + c
    fam["grid_axis"] = ax.reshape(n, 18)
    th = rng.uniform(-np.pi, np.pi, (n, 1))
    px, py = g[None, :, 0] * sx, g[None, :, 1] * sy
    rot = np.stack([np.cos(th) * px - np.sin(th) * py, np.sin(th) * px + np.cos(th) * py], 2) + c
    fam["grid_rot"] = rot.reshape(n, 18)
    k = rng.randint(1, 30, (n, 1)).astype(np.float64)
    ci = rng.randint(0, 1024, (n, 1, 2)).astype(np.float64)
    a, b = rng.randint(-6, 7, (n, 1)).astype(np.float64), rng.randint(-6, 7, (n, 1)).astype(np.float64)
    a[(a == 0) & (b == 0)] = 1
    lat = np.stack([g[None, :, 0] * a * k - g[None, :, 1] * b * k, g[None, :, 0] * b * k + g[None, :, 1] * a * k], 2) + ci
    fam["grid_int"] = lat.reshape(n, 18)
    m = max(n // 8, 1)
    t = rng.uniform(-30, 30, (m, 9, 1))
    d = rng.normal(size=(m, 1, 2))
    fam["collinear"] = (rng.uniform(0, 1024, (m, 1, 2)) + t * d).reshape(m, 18)
    dup = S.gen_pointsets(m, seed + 2).reshape(m, 9, 2)
    dup[:, 3:] = dup[:, rng.randint(0, 3, 6)]
    fam["duplicate"] = dup.reshape(m, 18)
    fam["tiny"] = (rng.uniform(0, 1024, (m, 1, 2)) + rng.normal(0, 1e-3, (m, 9, 2))).reshape(m, 18)
    return {k_: v.astype(np.float32) for k_, v in fam.items()}


def libm_check(dev):
    L = _lib.lib()
    libm = ctypes.CDLL("libm.so.6")
    hi = int(np.float32(4.0).view(np.uint32))
    res = {}
    harness = os.path.join(ROOT, "tests", "host_harness", "liblibm_host.so")
    H = ctypes.CDLL(harness) if os.path.exists(harness) else None
    for which, name in ((0, "cosf"), (1, "sinf")):
        bad = 0
        for neg in (0, 1):
            for lo in range(0, hi, 1 << 26):
                bits = np.arange(lo, min(hi, lo + (1 << 26)), dtype=np.uint32) | np.uint32(0x80000000 if neg else 0)
                x = bits.view(np.float32)
                xd = torch.from_numpy(x).to(dev)
                out = torch.empty_like(xd)
                rc = L.orp_libm_eval(_lib.ptr(xd), None, x.size, which, _lib.ptr(out), _lib.stream_of(xd))
                assert rc == 0
                got = out.cpu().numpy()
                want = np.empty_like(x)
                if H is not None:      # the g++ build of the same header was checked against libm on the CPU side
                    H.host_libm_eval(x.ctypes.data_as(ctypes.c_void_p), x.size, which, want.ctypes.data_as(ctypes.c_void_p))
                else:
                    want = (np.sin if which else np.cos)(x)
                bad += int(np.count_nonzero(got.view(np.uint32) != want.view(np.uint32)))
        res[name] = bad
    return res


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    dev = torch.device("cuda:0")
    print("device libm vs host (mismatching floats in (-4, 4)):", libm_check(dev), flush=True)
    total = bad_total = 0
    for name, pts in families(n).items():
        t0 = time.time()
        want = O.minarearect(pts)
        t1 = time.time()
        got = minaerarect(torch.from_numpy(pts).to(dev)).cpu().numpy()
        neq = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        rows = np.nonzero(neq.any(1))[0]
        rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        total += len(pts)
        bad_total += len(rows)
        print("%-10s sets %8d  differing %6d  (max rel diff %.3g)  oracle %.1fs" %
              (name, len(pts), len(rows), float(np.nanmax(rel)) if len(pts) else 0.0, t1 - t0), flush=True)
        for r in rows[:3]:
            print("   set", r, "pts", pts[r].tolist())
            print("       want", want[r].tolist())
            print("       got ", got[r].tolist())
    print("TOTAL sets %d differing %d" % (total, bad_total))


if __name__ == "__main__":
    main()
