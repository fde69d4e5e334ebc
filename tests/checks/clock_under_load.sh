#!/bin/bash
# Development aid (GPU box): engine clock and socket power while the DeformConv pair launch runs back to back, with and
# without the tap-granular split (ORP_DCN_KSPLIT=0), sampled with rocm-smi.
cd "$(dirname "$0")/../.."
cat > /tmp/dcn_loop.py <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from orientedreppoints_amd.mmdet_ops import deform_conv_forward_pair
dev = torch.device("cuda:0")
B = int(os.environ.get("LOOP_B", "1"))
sizes = (128, 64, 32, 16, 8)
fa = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
fb = [torch.randn(B, 256, n, n, device=dev).contiguous(memory_format=torch.channels_last) for n in sizes]
of = [torch.randn(B, 18, n, n, device=dev) * 2 for n in sizes]
w1, w2 = torch.randn(256, 256, 3, 3, device=dev) * 0.02, torch.randn(256, 256, 3, 3, device=dev) * 0.02
t0 = time.time(); n = 0
while time.time() - t0 < float(os.environ.get("LOOP_S", "6")):
    for _ in range(200):
        deform_conv_forward_pair(fa, fb, of, w1, w2, 1, 1, 1, relu=True)
    torch.cuda.synchronize(); n += 200
print("calls %d, %.1f us per call" % (n, (time.time() - t0) / n * 1e6))
PY
sample() { for i in 1 2 3; do sleep 1; /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Socket" | tr '\n' ' '; echo; done; }
echo "== idle"; /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
for ks in 0 1; do
  echo "== pair launch loop, B=1, ORP_DCN_KSPLIT=$ks"
  ORP_DCN_KSPLIT=$ks python /tmp/dcn_loop.py & pid=$!
  sleep 2.5; sample; wait $pid
done
echo "== pair launch loop, B=2, split"; LOOP_B=2 python /tmp/dcn_loop.py & pid=$!; sleep 2.5; sample; wait $pid
if [ -x tests/checks/mfma_rate ]; then echo "== mfma_rate microbenchmark"; (for i in 1 2 3 4 5 6; do tests/checks/mfma_rate > /tmp/mf.log; done) & pid=$!; sleep 1; sample; wait $pid; tail -2 /tmp/mf.log; fi
