#!/bin/bash
# Development aid (GPU box): engine clock and socket power while the DeformConv pair launch runs back to back in the exact-fp32
# mode (ORP_DCN_SPLIT=0) and on the bf16-split path (6 / 9 products), sampled with rocm-smi; then the bf16 MFMA microbenchmark.
cd "$(dirname "$0")/../.."
cat > /tmp/dcn_loop.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
dev = torch.device("cuda:0")
sizes = (128, 64, 32, 16, 8)
fa = [torch.randn(1, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
fb = [torch.randn(1, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
of = [torch.randn(1, 18, n, n, device=dev) * 2 for n in sizes]
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(200):
        deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); n += 200
print("calls %d, %.1f us per call" % (n, (time.time() - t0) / n * 1e6))
PY
sample() { for i in 1 2 3; do sleep 1; /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket" | tr '\n' ' '; echo; done; }
for m in 0 3 6 9; do
  echo "== pair launch loop, ORP_DCN_SPLIT=$m"
  ORP_DCN_SPLIT=$m python /tmp/dcn_loop.py & pid=$!
  sleep 2.5; sample; wait $pid
done
if [ -x tests/checks/mfma_rate_bf16 ]; then echo "== bf16 mfma microbenchmark"; (for i in 1 2 3 4 5 6 7 8 9 10; do tests/checks/mfma_rate_bf16 > /tmp/mf.log; done) & pid=$!; sleep 0.5; sample; wait $pid; head -3 /tmp/mf.log; fi
