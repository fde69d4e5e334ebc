"""Development aid (GPU box): the wave-specialised half-precision DeformConv forward under soak -- N launches (fp16 and bf16, one and two
images, five levels) next to a stream of library GEMMs and a second stream of the same kernel, EVERY result compared on the device with
the first one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
dev = torch.device("cuda:0")
N = int(os.environ.get("SOAK_N", "2000"))
total_bad = 0
for dt in (torch.float16, torch.bfloat16):
    for size, B in ((1024, 1), (512, 2)):
        torch.manual_seed(1)
        sizes = [size // s for s in (8, 16, 32, 64, 128)]
        w = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).to(dt)
        xs = [torch.randn(B, 256, n, n, device=dev).to(dt).contiguous(memory_format=torch.channels_last) for n in sizes]
        offs = [(torch.randn(B, 18, n, n, device=dev) * 2).to(dt) for n in sizes]
        ref = [o.clone() for o in deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)]
        a = torch.randn(2048, 2048, device=dev); b = torch.randn(2048, 2048, device=dev)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for it in range(N):
            with torch.cuda.stream(s1):
                a @ b
            with torch.cuda.stream(s2):
                deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
            outs = deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
            same = torch.stack([torch.equal(o, r) if False else (o == r).all() for o, r in zip(outs, ref)]).all()
            bad += (~same).to(torch.int64)
        torch.cuda.synchronize()
        print("%s %d^2 B=%d: %d launches next to a GEMM stream and a second stream of the same kernel, differing from the first: %d" % (str(dt).split('.')[-1], size, B, N, int(bad)))
        total_bad += int(bad)
print("TOTAL launches with a differing result: %d" % total_bad)
