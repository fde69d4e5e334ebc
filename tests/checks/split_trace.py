"""Development aid (GPU box): WHERE do the wrong rows of the MT = 1 DeformConv split launch come from?  Needs a library built with
-DORP_DCNS_TRACE=1|2 (tools/build_variant.py): every workgroup dumps its bilinear coefficient table (and, TRACE=2, a hash of every
A-tile row of every phase) into a device buffer.  The launch is repeated next to a GEMM stream and a stream of tower-convolution
launches; the trace of the first launch whose OUTPUT differs from the reference launch is kept on the device and compared with the
reference launch's trace: table entries / (phase, row) hashes that differ, next to the output positions that differ.

  ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace2.so NPROD=6 BATCH=1 N=3000 python tests/checks/split_trace.py
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi

dev = torch.device("cuda:0")
torch.manual_seed(0)
L = _lib.lib()
NPROD = int(os.environ.get("NPROD", "6"))
B = int(os.environ.get("BATCH", "1"))
N = int(os.environ.get("N", "3000"))
sizes = (32, 16, 8, 4, 2)
TAB0, ROW0 = 16, 16 + (1 << 20)
trace = torch.zeros(16 + (2 << 20), dtype=torch.int32, device=dev)
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
fa = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
fb = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
ca = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
xa = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
side, third = torch.cuda.Stream(), torch.cuda.Stream()
gemm_a = torch.randn(4096, 4096, device=dev)


def run():
    r = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=False)
    return list(r[0]) + list(r[1])


with torch.no_grad():
    L.orp_dcn_set_split_mode(0)
    exact = [t.clone() for t in run()]
    L.orp_dcn_set_split_mode(NPROD)
    L.orp_debug_amax_log(ctypes.c_void_p(trace.data_ptr()), 1 << 20)
    tols = [1e-4 * float(e.abs().max()) for e in exact]

    def bad_of(out):                                      # a launch is BAD when it is off the exact-fp32 result beyond 1e-4 of scale
        return torch.stack([((x - y).abs() > t).any() for x, y, t in zip(out, exact, tols)]).any()
    print("library %s; %d products, B=%d, levels %s" % (os.environ.get("ORP_HIP_LIB", "in-tree"), NPROD, B, sizes))
    nbad = torch.zeros((), dtype=torch.int64, device=dev)
    first_bad = torch.full((), -1, dtype=torch.int64, device=dev)
    first_good = torch.full((), -1, dtype=torch.int64, device=dev)
    snap = [torch.zeros_like(t) for t in exact]
    ref = [torch.zeros_like(t) for t in exact]
    snap_trace = torch.zeros_like(trace)
    ref_trace = torch.zeros_like(trace)
    table_ref = None
    if NPROD != 6:
        # the coefficient table does not depend on the arithmetic mode: a GOOD six-product launch's table is the reference
        L.orp_dcn_set_split_mode(6)
        for _ in range(50):
            out = run()
            torch.cuda.synchronize()
            if not bool(bad_of(out)):
                table_ref = trace.clone()
                break
        L.orp_dcn_set_split_mode(NPROD)
        print("reference table from a good six-product launch: %s" % (table_ref is not None))
    for i in range(N):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                gemm_a = (gemm_a @ gemm_a).clamp_(-1, 1)
        if i % 2 == 0:
            with torch.cuda.stream(third):
                conv_split_multi(xa, ca, xa, ca, nprod=6)
        out = run()
        flag = bad_of(out)
        is_first = flag & (first_bad < 0)
        is_good = (~flag) & (first_good < 0)
        first_bad = torch.where(is_first, torch.full_like(first_bad, i), first_bad)
        first_good = torch.where(is_good, torch.full_like(first_good, i), first_good)
        for s_, r_, o in zip(snap, ref, out):
            s_.copy_(torch.where(is_first, o, s_))
            r_.copy_(torch.where(is_good, o, r_))
        snap_trace.copy_(torch.where(is_first, trace, snap_trace))
        ref_trace.copy_(torch.where(is_good, trace, ref_trace))
        nbad += flag.to(torch.int64)
    first_it = first_bad
    torch.cuda.synchronize()
    L.orp_debug_amax_log(None, 0)
    print("%d launches, %d with an output off the exact-fp32 result beyond 1e-4 of scale; first bad launch %d, first good launch %d (the reference trace)"
          % (N, int(nbad), int(first_it), int(first_good)))
    if int(nbad):
        # tiles: MT = 1, 32 positions per tile, levels back to back
        tile0, t = [], 0
        for n in sizes:
            tile0.append(t); t += (B * n * n + 31) // 32
        total = t
        for k, (s_, r_) in enumerate(zip(snap, ref)):
            if not bool(((s_ - exact[k]).abs() > tols[k]).any()):
                continue
            ne = ((s_ - exact[k]).abs() > tols[k]).any(dim=1)
            idx = torch.nonzero(ne).tolist()
            lvl, conv = k % len(sizes), k // len(sizes)
            n = sizes[lvl]
            rows = sorted({(tile0[lvl] + (b * n * n + h * n + w) // 32, (b * n * n + h * n + w) % 32) for b, h, w in idx})
            print("  output tensor %d (layer %d, level %d): wrong positions -> (tile, row) %s" % (k, conv, lvl, rows[:24]))
        tab_r = (table_ref if table_ref is not None else ref_trace)[TAB0:TAB0 + 2 * total * 32 * 9 * 8].view(2, total, 32 * 9, 8)
        tab_s = snap_trace[TAB0:TAB0 + 2 * total * 32 * 9 * 8].view(2, total, 32 * 9, 8)
        d = torch.nonzero((tab_r != tab_s).any(dim=-1)).tolist()
        print("  coefficient-table entries that differ from the reference launch's: %d%s" % (
            len(d), "".join("\n    layer %d tile %d entry %d (row %d tap %d): %s vs reference %s" % (
                c, t_, e, e // 9, e % 9, [hex(v & 0xffffffff) for v in tab_s[c, t_, e].tolist()],
                [hex(v & 0xffffffff) for v in tab_r[c, t_, e].tolist()]) for c, t_, e in d[:40])))
        rows_r = ref_trace[ROW0:ROW0 + 2 * total * 128 * 32].view(2, total, 128, 32)
        rows_s = snap_trace[ROW0:ROW0 + 2 * total * 128 * 32].view(2, total, 128, 32)
        d = torch.nonzero(rows_r != rows_s).tolist()
        if int((rows_r != 0).sum()) == 0:
            print("  (no row hashes in this build: TRACE=1)")
        else:
            print("  A-tile row hashes that differ from the reference launch's: %d; (layer, tile, phase, row): %s" % (len(d), d[:60]))
