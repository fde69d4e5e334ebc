"""Candidate selection (class max + per-level top-2000): orp_pp_select vs the torch max + topk route, event-timed.
    python tests/checks/time_select.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from orientedreppoints_amd.mmdet_models.core import select_candidates  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
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
    for sizes in ([16384, 4096, 1024, 256, 64], [36864, 9216, 2304, 576, 144]):
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        sig = torch.sigmoid(torch.randn(15, int(offs[-1]), device=dev) * 2 - 3)

        def torch_route():
            mx = sig.max(dim=0)[0]
            out = []
            for l, n_l in enumerate(sizes):
                if n_l > 2000:
                    out.append(mx[int(offs[l]):int(offs[l + 1])].topk(2000)[1] + int(offs[l]))
                else:
                    out.append(torch.arange(int(offs[l]), int(offs[l + 1]), device=dev))
            return torch.cat(out)
        a = select_candidates(sig, offs, 2000)
        assert torch.equal(a, torch_route())
        print("levels %s: orp_pp_select %.1f us   torch max + topk %.1f us" % (sizes, timed(lambda: select_candidates(sig, offs, 2000)), timed(torch_route)))


if __name__ == '__main__':
    main()
