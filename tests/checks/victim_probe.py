"""Development aid (GPU box): does the MT = 1 split kernel disturb OTHER kernels' waves on the CUs it shares with them?
The victim is tests/checks/sgpr_mask_probe.hip's kernel (compare / mask / select code checked per lane against a select-free
evaluation of the same values; 0 errors in 2e11 evaluations on its own), built as libvictim.so; the aggressor runs on a second stream:
the library's tower-convolution pair launch on a small pyramid (tile height 1), on a large one (tile height 3), the DeformConv pair
launch, or a library GEMM.  AGGR=conv_small|conv_big|dcn_small|gemm|none  N=400 python tests/checks/victim_probe.py"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from orientedreppoints_amd import _lib
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi

dev = torch.device("cuda:0")
V = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvictim.so"))
V.victim_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
N = int(os.environ.get("N", "400"))
ITERS = int(os.environ.get("VICTIM_ITERS", "2000"))
torch.manual_seed(0)
ca = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
cb = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)


def cl(B, sizes):
    return [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]


small, big = (32, 16, 8, 4, 2), (128, 64, 32, 16, 8)
xs, xb_ = cl(2, small), cl(2, small)
xL, xLb = cl(1, big), cl(1, big)
fa = [torch.randn(2, 256, n, n, device=dev) for n in small]
of = [torch.randn(2, 18, n, n, device=dev) * 2 for n in small]
w1 = torch.randn(256, 256, 3, 3, device=dev) * 0.02
g = torch.randn(2048, 2048, device=dev)
aggr_stream, vict_stream = torch.cuda.Stream(), torch.cuda.Stream()
AGGRS = os.environ.get("AGGR", "none,gemm,conv_big,conv_small,dcn_small").split(",")
print("library %s (%s)" % (_lib.lib().orp_version().decode(), os.environ.get("ORP_HIP_LIB", "in-tree")))
with torch.no_grad():
    for aggr in AGGRS:
        bad = torch.zeros(256, dtype=torch.int32, device=dev)
        sink = torch.zeros(16, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        for i in range(N):
            with torch.cuda.stream(aggr_stream):
                for _ in range(4):
                    if aggr == "gemm":
                        g = (g @ g).clamp_(-1, 1)
                    elif aggr == "conv_small":
                        conv_split_multi(xs, ca, xb_, cb, nprod=6)
                    elif aggr == "conv_big":
                        conv_split_multi(xL, ca, xLb, cb, nprod=6)
                    elif aggr == "dcn_small":
                        deform_conv_forward_pair(fa, fa, of, w1, w1, 1, 1, 1, relu=True)
            rc = V.victim_launch(512, ITERS, ctypes.c_void_p(bad.data_ptr()), ctypes.c_void_p(sink.data_ptr()), 0,
                                 ctypes.c_void_p(vict_stream.cuda_stream))
            assert rc == 0
        torch.cuda.synchronize()
        h = bad.view(64, 4).cpu()
        tot = h.sum(0).tolist()
        q = h.view(4, 16, 4).sum(1).tolist()
        print("aggressor %-10s: %d victim launches (%.1e evaluations per weight) in %.1f s: wrong w.x %d w.y %d w.z %d w.w %d; by lane quarter %s" % (
            aggr, N, N * 512 * 512 * ITERS, time.time() - t0, tot[0], tot[1], tot[2], tot[3], q))
