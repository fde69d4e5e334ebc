"""soft_rnms (CPU-only op, host function of liborp_hip.so) against golden vectors produced by the reference's own
rnms_cpu.cpp compiled unmodified (tests/golden/make_golden_softnms.py).  No GPU involved: the reference op is CPU only."""
import os

import numpy as np
import pytest

from orientedreppoints_amd.mmdet_ops.nms_wrapper import soft_rnms

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "soft_rnms.npz"))
METHODS = {0: 'original', 1: 'linear', 2: 'gaussian'}


@pytest.mark.parametrize("case", [str(c) for c in G["cases"]])
def test_soft_rnms_matches_reference(case):
    name, m, t, s, ms = case.rsplit("_", 4)
    method, thr, sigma, min_score = int(m[1:]), float(t[1:]), float(s[1:]), float(ms[2:])
    dets = G["in_" + name]
    want = G["out_" + case]
    new_dets, inds = soft_rnms(dets, thr, METHODS[method], sigma, min_score)
    assert np.array_equal(inds, want[:, 9].astype(np.int64))            # same survivors, same order
    assert np.array_equal(new_dets[:, :8], want[:, :8])
    if method == 2:
        assert np.max(np.abs(new_dets[:, 8] - want[:, 8])) <= 1e-6      # expf vs std::exp
    else:
        assert np.array_equal(new_dets[:, 8], want[:, 8])               # + - * / only: bit-exact


def test_soft_rnms_api_edges():
    import torch
    d = G["in_uniform200"]
    a, ia = soft_rnms(torch.from_numpy(d), 0.3)
    b, ib = soft_rnms(d, 0.3)
    assert isinstance(a, torch.Tensor) and ia.dtype == torch.long and np.array_equal(a.numpy(), b) and np.array_equal(ia.numpy(), ib)
    with pytest.raises(ValueError):
        soft_rnms(d, 0.3, method='nope')
    with pytest.raises(TypeError):
        soft_rnms([1, 2, 3], 0.3)
    e, ie = soft_rnms(np.zeros((0, 9), np.float32), 0.3)
    assert e.shape == (0, 9) and ie.shape == (0,)
