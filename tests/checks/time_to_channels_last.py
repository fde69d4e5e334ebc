"""Dev script (GPU): the NCHW -> channels-last transposition of the five pyramid levels (256 channels; with the range words, as the
training nodes and the inference head call it), ORP_TOCL_WIDE=0 (32 x 32 tiles, 4-byte accesses) vs 1 (64 x 64 tiles, 16-byte accesses).
   ORP_TOCL_WIDE=0|1 python tests/checks/time_to_channels_last.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops.fused_norm import to_channels_last_multi
dev = torch.device("cuda:0")
for B, size in ((1, 1024), (2, 1024), (1, 1536)):
    xs = [torch.randn(B, 256, size // s, size // s, device=dev) for s in (8, 16, 32, 64, 128)]
    for _ in range(5):
        to_channels_last_multi(xs, amax_slots=[0] * 5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        to_channels_last_multi(xs, amax_slots=[0] * 5)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    mb = sum(x.numel() for x in xs) * 8 / 1e6
    print("ORP_TOCL_WIDE=%s  B=%d %d^2: %.1f us per call (transposition + range launch), %.1f MB moved = %.2f TB/s"
          % (os.environ.get("ORP_TOCL_WIDE", "1"), B, size, us, mb, mb / us / 1e6 * 1e6 / 1e6), flush=True)
