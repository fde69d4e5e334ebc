#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
(cd tests/checks && timeout 300 ./mfma_refill_victim 200) > $OUT/r05_mfma_refill_victim.log 2>&1
cat $OUT/r05_mfma_refill_victim.log | cut -c1-330
ORP_HIP_LIB=build_variants/liborp_hip_d0lag.so AGGR=conv_small,dcn_small N=300 timeout 300 python tests/checks/victim_probe.py 2>&1 | grep "aggressor\|library" | cut -c1-330 | tee $OUT/r05_victim_lag.log
