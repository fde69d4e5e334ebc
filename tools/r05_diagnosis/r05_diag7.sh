#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
N=300 timeout 500 python tests/checks/victim_probe.py > $OUT/r05_victim_default.log 2>&1
ORP_HIP_LIB=build_variants/liborp_hip_drain0.so AGGR=conv_small,dcn_small N=300 timeout 500 python tests/checks/victim_probe.py > $OUT/r05_victim_drain0.log 2>&1
cat $OUT/r05_victim_default.log $OUT/r05_victim_drain0.log | grep -v amdgpu.ids
