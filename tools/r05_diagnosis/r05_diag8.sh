#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
: > $OUT/r05_victim_bisect.log
for d in 1 2 4 8 16 32; do
  echo "== aggressor built with -DORP_DCNS_DRAIN=0 -DORP_DCNS_DBG=$d (1 no gathers, 2 no weight refills, 4 no MFMA, 8 no combine / LDS writes, 16 no A-fragment LDS reads, 32 no per-phase barrier)" >> $OUT/r05_victim_bisect.log
  ORP_HIP_LIB=build_variants/liborp_hip_d0dbg$d.so AGGR=conv_small,dcn_small N=200 timeout 300 python tests/checks/victim_probe.py 2>&1 | grep aggressor >> $OUT/r05_victim_bisect.log
done
echo "== aggressor: in-tree library, tile height forced to 3 / one workgroup per CU" >> $OUT/r05_victim_bisect.log
ORP_DCNS_MT=3 AGGR=conv_small N=200 timeout 300 python tests/checks/victim_probe.py 2>&1 | grep aggressor >> $OUT/r05_victim_bisect.log
ORP_DCNS_PAD_LDS=84 AGGR=conv_small N=200 timeout 300 python tests/checks/victim_probe.py 2>&1 | grep aggressor >> $OUT/r05_victim_bisect.log
cut -c1-260 $OUT/r05_victim_bisect.log
