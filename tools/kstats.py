#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 `--kernel-trace --stats --output-format csv` output dir.
usage: tools/kstats.py gpurun_out/<dir> [n]"""
import csv, glob, os, sys
src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
stats = glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True)[0]
for r in list(csv.DictReader(open(stats)))[:n]:
    print("%-110s calls %5s avg %10.1f us total %10.1f us %5s%%" % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3,
                                                                  float(r['TotalDurationNs']) / 1e3, r['Percentage']))
