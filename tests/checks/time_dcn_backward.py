"""Timing of the DeformConv backward at the training configuration (2 images x 1024^2: levels 128^2 .. 8^2, 256 -> 256):
MFMA implicit GEMMs (one call for the five levels) vs the column formulation (per level), HIP-event timed.
    python tests/checks/time_dcn_backward.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from orientedreppoints_amd.mmdet_ops import deform_conv_backward as bw  # noqa: E402


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    B = int(os.environ.get('B', 2))
    std = float(os.environ.get('OFF_STD', 2.0))
    xs = [torch.randn(B, 256, s, s, device=dev) for s in (128, 64, 32, 16, 8)]
    offs = [torch.randn(B, 18, s, s, device=dev) * std for s in (128, 64, 32, 16, 8)]
    gos = [torch.randn(B, 256, s, s, device=dev) for s in (128, 64, 32, 16, 8)]
    positives = int(os.environ.get('POSITIVES', 0))          # > 0: only that many clustered positions per level keep a gradient
    if positives:
        for g in gos:
            s_ = g.size(2)
            keep = torch.zeros(B, 1, s_, s_, device=dev)
            n_obj = max(1, min(positives, s_ * s_) // 9)
            cy = torch.randint(1, max(2, s_ - 1), (B, n_obj))
            cx = torch.randint(1, max(2, s_ - 1), (B, n_obj))
            for b in range(B):
                for y, x_ in zip(cy[b].tolist(), cx[b].tolist()):
                    keep[b, 0, max(0, y - 1):y + 2, max(0, x_ - 1):x_ + 2] = 1.0
            g.mul_(keep)
    w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
    args = ((1, 1), (1, 1), (1, 1))
    t_all = timed(lambda: bw.backward_mfma(xs, offs, w, gos, *args))
    # the library's own HIP events around the parts of one call: GEMM kernel (grad_input rows + grad_offset), sort + region
    # scatter, weight-gradient kernel + reduction
    import ctypes
    from orientedreppoints_amd import _lib
    L = _lib.lib()

    def prof(slot):
        tot = ctypes.c_double(0); cnt = ctypes.c_int(0)
        L.orp_profile_read(slot, ctypes.cast(ctypes.byref(tot), ctypes.c_void_p), ctypes.cast(ctypes.byref(cnt), ctypes.c_void_p), 1)
        return tot.value * 1e3
    L.orp_profile_enable(1)
    for sl in (8, 9, 10, 11):
        prof(sl)
    for _ in range(10):
        bw.backward_mfma(xs, offs, w, gos, *args)
    torch.cuda.synchronize()
    parts = [prof(sl) / 10 for sl in (8, 9, 10, 11)]
    L.orp_profile_enable(0)
    print("  per call (HIP events inside the library): whole %.1f us | input GEMM kernel %.1f us (%.1f TF/s) | bin + sort + bounds + descriptors + scatter %.1f us | "
          "weight kernel + reduction %.1f us" % (parts[0], parts[1], 2.0 * sum(B * s * s for s in (128, 64, 32, 16, 8)) * 2304 * 256 / parts[1] * 1e-6, parts[2], parts[3]))
    t_in = timed(lambda: bw.backward_mfma(xs, offs, w, gos, *args, need_input=True, need_weight=False))
    t_w = timed(lambda: bw.backward_mfma(xs, offs, w, gos, *args, need_input=False, need_weight=True))
    bw.USE_MFMA = False

    def column():
        for x, o, g in zip(xs, offs, gos):
            bw.backward_input(x, o, w, g, *args, 1, 1)
            bw.backward_parameters(x, o, w, g, *args, 1, 1)
    t_col = timed(column)
    npos = sum(B * s * s for s in (128, 64, 32, 16, 8))
    fl = 2.0 * npos * 2304 * 256
    print("positions %d  mfma all %.1f us (input %.1f us = %.1f TF/s, weight %.1f us = %.1f TF/s)   column formulation %.1f us"
          % (npos, t_all, t_in, fl / t_in * 1e-6, t_w, fl / t_w * 1e-6, t_col))


if __name__ == '__main__':
    main()
