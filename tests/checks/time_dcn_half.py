"""Development aid (GPU box): the half-precision DeformConv forward (fp16 / bf16), all five 1024^2 levels in one launch, symmetric
(ORP_DCNH_WS=0) against wave-specialised kernel (default), with a hash of the outputs (the two kernels must agree bit for bit)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_multi
dev = torch.device("cuda:0")
tag = "ORP_DCNH_WS=%s" % os.environ.get("ORP_DCNH_WS", "default")
for dt in (torch.float16, torch.bfloat16):
    for size, B in ((1024, 1), (1024, 2), (256, 2)):
        torch.manual_seed(0)
        sizes = [size // s for s in (8, 16, 32, 64, 128)]
        w = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).to(dt)
        xs = [torch.randn(B, 256, n, n, device=dev).to(dt).contiguous(memory_format=torch.channels_last) for n in sizes]
        offs = [(torch.randn(B, 18, n, n, device=dev) * 2).to(dt) for n in sizes]
        outs = deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
        h = hashlib.sha1(b"".join(o.float().cpu().numpy().tobytes() for o in outs)).hexdigest()[:12]
        for _ in range(10):
            deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            deform_conv_forward_multi(xs, offs, w, 1, 1, 1, relu=True)
        e1.record(); torch.cuda.synchronize()
        print("[%s] %s %d^2 B=%d: %.1f us per call, outputs sha1 %s" % (tag, str(dt).split('.')[-1], size, B, e0.elapsed_time(e1) / 40 * 1e3, h))
