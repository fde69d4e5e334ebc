"""Development aid (GPU box; run under `timeout`): the split kernel family under soak with EVERY result compared (round 4's soaks
compared one result in 20 / 50).  Tower-convolution pair launches (PLAIN instantiation) and DeformConv pair launches, tile heights
1 (small pyramids: two workgroups per CU) and 3, one and two images, next to a stream of library GEMMs and a second stream running the
same kernel; the comparison runs on the device (no host synchronisation inside the loop), the first mismatching result of a
configuration is kept and described (which positions, how many channels).

  SOAK_N=2000 KINDS=conv,dcn NPRODS=6,3 python tests/checks/soak_split_full.py
  ORP_HIP_LIB=build_variants/liborp_hip_drain0.so ... (a library built with -DORP_DCNS_DRAIN=0)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi

dev = torch.device("cuda:0")
torch.manual_seed(0)
N = int(os.environ.get("SOAK_N", "2000"))
KINDS = os.environ.get("KINDS", "conv,dcn").split(",")
NPRODS = [int(v) for v in os.environ.get("NPRODS", "6,3").split(",")]
NEIGHBOURS = os.environ.get("NEIGHBOURS", "1") == "1"
L = _lib.lib()
print("library %s (%s), ORP_DCNS_MT=%s ORP_DCNS_PAD_LDS=%s, %d launches per configuration, every result compared" % (
    L.orp_version().decode(), os.environ.get("ORP_HIP_LIB", "in-tree"), os.environ.get("ORP_DCNS_MT"), os.environ.get("ORP_DCNS_PAD_LDS"), N))

ca = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
cb = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
side, third = torch.cuda.Stream(), torch.cuda.Stream()
gemm_a = torch.randn(4096, 4096, device=dev)


def describe(a, b):
    ne = a != b
    pos = ne.any(dim=1)
    ch = ne.sum(dim=1)[pos]
    d = torch.nan_to_num((a - b).abs())
    return "%s: %d elements differ, max |diff| %.3e (scale %.2e), nan %d; positions touched %d of %d, channels per touched position min %d max %d, first (b,h,w) %s" % (
        tuple(a.shape), int(ne.sum()), float(d.max()), float(b.abs().max()), int(torch.isnan(a).sum()), int(pos.sum()), pos.numel(),
        int(ch.min()), int(ch.max()), torch.nonzero(pos)[:5].tolist())


total_bad = 0
with torch.no_grad():
    for kind in KINDS:
        for nprod in NPRODS:
            for B, sizes in ((1, (32, 16, 8, 4, 2)), (2, (32, 16, 8, 4, 2)), (2, (64, 32, 16, 8, 4)), (1, (128, 64, 32, 16, 8))):
                big = sizes[0] >= 128
                n_it = N // 4 if big else N
                if kind == 'conv':
                    xa = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
                    xb = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]

                    def run(swap=False):
                        r = conv_split_multi(xb, cb, xa, ca, nprod=nprod) if swap else conv_split_multi(xa, ca, xb, cb, nprod=nprod)
                        return list(r[0]) + list(r[1])
                else:
                    L.orp_dcn_set_split_mode(nprod)
                    fa = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
                    fb = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
                    of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]

                    def run(swap=False):
                        r = deform_conv_forward_pair(fb, fa, of, w2, w1, 1, 1, 1, relu=True) if swap else \
                            deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
                        return list(r[0]) + list(r[1])
                ref = [t.clone() for t in run()]
                nbad = torch.zeros((), dtype=torch.int64, device=dev)
                first_it = torch.full((), -1, dtype=torch.int64, device=dev)
                snap = [torch.zeros_like(t) for t in ref] if not big else None
                t0 = time.time()
                for i in range(n_it):
                    if NEIGHBOURS and i % 4 == 0:
                        with torch.cuda.stream(side):
                            gemm_a = (gemm_a @ gemm_a).clamp_(-1, 1)
                    if NEIGHBOURS and i % 2 == 0:
                        with torch.cuda.stream(third):
                            run(swap=True)
                    out = run()
                    flag = torch.stack([(x != y).any() for x, y in zip(out, ref)]).any()
                    is_first = flag & (first_it < 0)
                    first_it = torch.where(is_first, torch.full_like(first_it, i), first_it)
                    if snap is not None:
                        for s_, o in zip(snap, out):
                            s_.copy_(torch.where(is_first, o, s_))
                    nbad += flag.to(torch.int64)
                torch.cuda.synchronize()
                nb = int(nbad)
                total_bad += nb
                print("%s pair, %d products, B=%d, levels %s: %d launches%s in %.1f s, launches with a result that differs from the first one: %d"
                      % (kind, nprod, B, sizes, n_it, " next to a GEMM stream and a second stream of the same kernel" if NEIGHBOURS else "",
                         time.time() - t0, nb))
                if nb and snap is not None:
                    print("    first mismatch at launch %d:" % int(first_it))
                    for k, (s_, r_) in enumerate(zip(snap, ref)):
                        if not torch.equal(s_, r_):
                            print("      tensor %d %s" % (k, describe(s_, r_)))
    L.orp_dcn_set_split_mode(-1)
print("TOTAL launches with a differing result: %d" % total_bad)
