#!/bin/bash
# usage: tools/pmc_pass.sh <outdir-under-gpurun_out> <which> "<counters>"   (one rocprofv3 --pmc pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc $3 --output-format csv -d $R/gpurun_out/$1 -- python $R/tools/run_hot_kernels.py $2 5 > $R/gpurun_out/$1.log 2>&1
cd $R
python - "$1" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob('gpurun_out/%s/*/*counter_collection.csv' % d)
if not f:
    print('no counter csv', glob.glob('gpurun_out/%s/*/*' % d)); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name'][:60]
    if 'dcn_fwd' in k or 'conv_split' in k or 'dcn_bwd' in k or 'nms_' in k or 'convex' in k or 'minarearect' in k or 'chamfer' in k or 'assign' in k:
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    print(k, {n: sum(v) / len(v) for n, v in c.items()}, 'launches', max(len(v) for v in c.values()))
PY
