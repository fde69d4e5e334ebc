"""Development aid (GPU box; run under `timeout`): the tap-granular split of the DeformConv pair launch under soak -- thousands
of back-to-back launches (2 and 3 images: 456 / 683 tiles) while a second stream keeps the CUs busy with library GEMMs, every
result compared bit for bit with the first one.  A hand-over that could hang or race would show here."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
from orientedreppoints_amd import _lib
POISON = os.environ.get("SOAK_POISON", "1") == "1"

dev = torch.device("cuda:0")
torch.manual_seed(0)
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device=dev)
N = int(os.environ.get("SOAK_N", "1500"))
for B in (2, 3):
    sizes = (128, 64, 32, 16, 8)
    fa = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    fb = [torch.randn(B, 256, n, n, device=dev) for n in sizes]
    of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
    mode = _lib.lib().orp_dcn_get_split_mode()               # ORP_DCN_SPLIT: 0 = exact fp32 MFMA (tap-granular split), 6 / 9 = bf16-split products
    _lib.lib().orp_dcn_set_split_mode(0)
    exact = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    exact = [t.clone() for t in exact[0] + exact[1]]
    _lib.lib().orp_dcn_set_split_mode(mode)
    ref = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    ref = [t.clone() for t in ref[0] + ref[1]]
    dev_rel = max(float((x - y).abs().max()) / float(y.abs().max()) for x, y in zip(ref, exact))
    print("B=%d: arithmetic mode %d, first result vs the exact-fp32 launch: max |diff| / max |out| = %.2e" % (B, mode, dev_rel))
    assert dev_rel <= 1e-5
    bad = 0
    t0 = time.time()
    for i in range(N):
        if i % 4 == 0:
            with torch.cuda.stream(side):
                a = (a @ a).clamp_(-1, 1)                       # keeps CUs / LDS busy next to the split launch
        if POISON:
            _lib.workspace(dev, 1).fill_(0xFF)                 # NaN patterns in the reused scratch images
        out = deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
        if i % 50 == 0 or i == N - 1:
            bad += sum(0 if torch.equal(x, y) else 1 for x, y in zip(out[0] + out[1], ref))
    torch.cuda.synchronize()
    print("B=%d: %d split launches next to a GEMM stream in %.1f s, mismatching tensors in the sampled results: %d" % (B, N, time.time() - t0, bad))
