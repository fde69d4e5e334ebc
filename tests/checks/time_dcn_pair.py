"""Development aid (GPU box): the head's DeformConv pair launch at the BASELINE shapes, timed by rocprofv3-independent
torch events around back-to-back launches.  ORP_DCN_KSPLIT=0 switches the tap-granular split off for comparison."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair

from orientedreppoints_amd import _lib
dev = torch.device("cuda:0")
MODE = _lib.lib().orp_dcn_get_split_mode()        # ORP_DCN_SPLIT = 0 (exact fp32 MFMA) | 6 | 9 (bf16-split products) | 3 (two fp16 pieces)
torch.manual_seed(0)
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
for size, B in ((1024, 1), (1024, 2), (1536, 1)):
    sizes = [size // s for s in (8, 16, 32, 64, 128)]
    fa = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    fb = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
    for fmt, name in ((torch.contiguous_format, 'nchw'), (torch.channels_last, 'nhwc')):
        a = [x.contiguous(memory_format=fmt) for x in fa]
        b = [x.contiguous(memory_format=fmt) for x in fb]
        for _ in range(5):
            deform_conv_forward_pair(a, b, of, w1, w2, 1, 1, 1, relu=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 30
        e0.record()
        for _ in range(n):
            deform_conv_forward_pair(a, b, of, w1, w2, 1, 1, 1, relu=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        flop = 2 * 2 * B * sum(s * s for s in sizes) * 256 * 256 * 9
        print("pair %4d^2 B=%d %s: %.1f us per call (incl. transposition for nchw), %.1f TFLOP/s = %.3f of 157.3  [KSPLIT env %s, split mode %d]"
              % (size, B, name, us, flop / us / 1e6, flop / us / 1e6 / 157.3, os.environ.get('ORP_DCN_KSPLIT', 'default'), MODE))
