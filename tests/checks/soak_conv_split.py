"""Development aid (GPU box; run under `timeout`): the tower convolutions (orp_conv_split_multi pair launches, PLAIN instantiation
with the side accumulators) under soak -- back-to-back launches at several sizes (tile heights 1 and 3) while a second stream
keeps the matrix pipes busy with library GEMMs and a third runs the same convolution, every result compared bit for bit with
the first one.  The condition under which in-place register refills overtook queued MFMAs before the accumulator drain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops.fused_norm import conv_split_multi
from orientedreppoints_amd import _lib

dev = torch.device("cuda:0")
torch.manual_seed(0)
ca = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
cb = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
side, third = torch.cuda.Stream(), torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
N = int(os.environ.get("SOAK_N", "600"))
with torch.no_grad():
    for nprod in (6, 3):
        for B, sizes in ((1, (32, 16, 8, 4, 2)), (2, (32, 16, 8, 4, 2)), (1, (128, 64, 32, 16, 8)), (2, (128, 64, 32, 16, 8))):
            xa = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
            xb = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
            ref = conv_split_multi(xa, ca, xb, cb, nprod=nprod)
            ref = [t.clone() for t in ref[0] + ref[1]]
            bad = 0
            t0 = time.time()
            for i in range(N):
                if i % 4 == 0:
                    with torch.cuda.stream(side):
                        a = (a @ a).clamp_(-1, 1)
                if i % 2 == 0:
                    with torch.cuda.stream(third):
                        conv_split_multi(xb, cb, xa, ca, nprod=nprod)
                out = conv_split_multi(xa, ca, xb, cb, nprod=nprod)
                if i % 20 == 0 or i == N - 1:
                    bad += sum(0 if torch.equal(x, y) else 1 for x, y in zip(out[0] + out[1], ref))
            torch.cuda.synchronize()
            print("%d products, B=%d, levels %s: %d pair launches next to a GEMM stream and a second convolution stream in %.1f s, "
                  "mismatching tensors in the sampled results: %d" % (nprod, B, sizes, N, time.time() - t0, bad))
