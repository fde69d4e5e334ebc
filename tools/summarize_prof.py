#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` output dir into the small summary committed under
profiles/:  <name>_kernel_stats.csv (top kernels) + <name>_step.txt (per-step breakdown between two nms_mask launches).
usage: tools/summarize_prof.py gpurun_out/<dir> profiles/<name> [step_index]"""
import collections
import csv
import glob
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
step = int(sys.argv[3]) if len(sys.argv) > 3 else 14
stats = glob.glob(os.path.join(src, '*', '*kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(stats)))
with open(dst + '_kernel_stats.csv', 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows[:60]:
        w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
trace = glob.glob(os.path.join(src, '*', '*kernel_trace.csv'))[0]
tr = list(csv.DictReader(open(trace)))
tr.sort(key=lambda r: int(r['Start_Timestamp']))
# one mask launch per bench step: the capacity-launch kernel of the sync-free path if present, else the exact one
marker = 'nms_mask_loop_kernel' if any('nms_mask_loop_kernel' in r['Kernel_Name'] for r in tr) else 'nms_mask_kernel'
idx = [i for i, r in enumerate(tr) if marker in r['Kernel_Name']]
with open(dst + '_step.txt', 'w') as f:
    if len(idx) > step + 1:
        a, b = idx[step], idx[step + 1]
        seg = tr[a + 1:b + 1]
        t0, t1 = int(tr[a]['End_Timestamp']), int(tr[b]['End_Timestamp'])
        agg = collections.defaultdict(lambda: [0, 0, 0])
        busy = 0
        for r in seg:
            d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
            busy += d
            k = r['Kernel_Name'][:110]
            agg[k][0] += d; agg[k][1] += 1; agg[k][2] = r['VGPR_Count'] + '/' + r['Accum_VGPR_Count'] + ' lds ' + r['LDS_Block_Size']
        f.write('one bench step (between nms_mask launch %d and %d): wall %.3f ms (under tracing), kernel busy %.3f ms, %d kernels\n'
                % (step, step + 1, (t1 - t0) / 1e6, busy / 1e6, len(seg)))
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
            f.write('%-112s calls %3d  total %9.1f us  vgpr/agpr %s\n' % (k, v[1], v[0] / 1e3, v[2]))
print(open(dst + '_step.txt').read()[:600])
