"""Development aid (GPU box): the DeformConv weight gradient on the 16-bit pipe (dcn_bwd_weight16_kernel) under soak -- N calls at the
BASELINE training shapes (five levels, two images; a dense gradient and a sparse one: 40 non-zero positions) next to a stream of library
GEMMs and a second stream running the same backward, EVERY grad_weight compared on the device with the first one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw
dev = torch.device("cuda:0")
N = int(os.environ.get("SOAK_N", "1000"))
total_bad = 0
for size, B, sparse in ((1024, 2, False), (1024, 2, True), (512, 1, False)):
    torch.manual_seed(3)
    sizes = [size // s for s in (8, 16, 32, 64, 128)]
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.02
    xs = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    offs = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
    gos = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    if sparse:
        for g in gos:
            keep = torch.zeros(B, 1, g.size(2), g.size(3), device=dev)
            idx = torch.randint(0, keep.numel(), (8,), device=dev)
            keep.view(-1)[idx] = 1.0
            g.mul_(keep)
    call = lambda: bw.backward_mfma(xs, offs, w, gos, (1, 1), (1, 1), (1, 1), need_input=False)[2]   # noqa: E731
    ref = call().clone()
    a = torch.randn(2048, 2048, device=dev); b = torch.randn(2048, 2048, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    for it in range(N):
        with torch.cuda.stream(s1):
            a @ b
        with torch.cuda.stream(s2):
            call()
        bad += (~(call() == ref).all()).to(torch.int64)
    torch.cuda.synchronize()
    print("%d^2 B=%d %s gradient: %d calls next to a GEMM stream and a second stream of the same backward, grad_weight differing from the first: %d"
          % (size, B, "sparse" if sparse else "dense", N, int(bad)), flush=True)
    total_bad += int(bad)
print("TOTAL calls with a differing result: %d" % total_bad)
