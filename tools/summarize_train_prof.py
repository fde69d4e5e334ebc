#!/usr/bin/env python3
"""rocprofv3 `--kernel-trace --output-format csv` of `bench.py --mode train` -> per-kernel breakdown of ONE training step
(between two apaa_select launches `sel` launches apart: since round 3 the selection kernel runs ONCE per step for all
images, before that once per image).
usage: tools/summarize_train_prof.py gpurun_out/<dir> profiles/<name>_train_step.txt [select_launches_per_step=1] [step_index=6]"""
import collections
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
imgs = int(sys.argv[3]) if len(sys.argv) > 3 else 1
step = int(sys.argv[4]) if len(sys.argv) > 4 else 6
trace = glob.glob(os.path.join(src, '*', '*kernel_trace.csv'))[0]
tr = list(csv.DictReader(open(trace)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(tr) if 'apaa_select' in r['Kernel_Name']]
a, b = idx[step * imgs], idx[(step + 1) * imgs]
seg = tr[a + 1:b + 1]
t0, t1 = int(tr[a]['End_Timestamp']), int(tr[b]['End_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in seg:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    busy += d
    k = r['Kernel_Name'][:100]
    agg[k][0] += d
    agg[k][1] += 1
with open(dst, 'w') as f:
    f.write('one training step of bench.py --mode train (configs[2]: 2 x 1024^2 images, 64 gts each, APAA on; between two '
            'apaa_select launches %d apart): wall %.3f ms (under tracing), kernel busy %.3f ms, %d kernels\n'
            % (imgs, (t1 - t0) / 1e6, busy / 1e6, len(seg)))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        f.write('%-100s calls %4d  total %9.1f us\n' % (k, v[1], v[0] / 1e3))
print(open(dst).read()[:6000])
