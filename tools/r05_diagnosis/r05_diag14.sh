#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
run_gb() { tag=$1; shift; env "$@" timeout 300 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; echo "$tag: $(tail -1 $OUT/r05_gb_$tag.log | cut -c90-500)"; }
run_gb own_drain2 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on
run_gb own_fence SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_fence.so
run_gb own_drain0 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_drain0.so
run_gb noown_drain2 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_noown.so
run_gb own_drain2_m3 SIZE=256 BATCH=2 DEPTH=3 ITERS=1000 MODE=3 SPLIT=on
AGGR=conv_small,dcn_small N=300 timeout 200 python tests/checks/victim_probe.py 2>&1 | grep "aggressor\|library" | cut -c1-300 | tee $OUT/r05_victim_own.log
run_soak() { tag=$1; shift; env "$@" SOAK_N=2000 timeout 400 python tests/checks/soak_split_full.py > $OUT/r05_soak_$tag.log 2>&1; echo "$tag: $(tail -1 $OUT/r05_soak_$tag.log)"; }
run_soak own_drain2
run_soak own_fence ORP_HIP_LIB=build_variants/liborp_hip_fence.so
run_soak own_drain0 ORP_HIP_LIB=build_variants/liborp_hip_drain0.so
