#!/bin/bash
# Round-5 diagnosis pass 1 (GPU box): MFMA source-register hazard probe, fully compared kernel soak (default and -DORP_DCNS_DRAIN=0
# builds), stage-by-stage bitwise comparison of graph replays with the eager step (six-product and fp16-pieces arithmetic).
OUT=gpurun_out
mkdir -p $OUT
(cd tests/checks && timeout 300 ./mfma_war 200 2000) > $OUT/r05_mfma_war.log 2>&1
SOAK_N=1500 timeout 500 python tests/checks/soak_split_full.py > $OUT/r05_soak_full_default.log 2>&1
ORP_HIP_LIB=build_variants/liborp_hip_drain0.so SOAK_N=1500 timeout 500 python tests/checks/soak_split_full.py > $OUT/r05_soak_full_drain0.log 2>&1
run_gb() { tag=$1; shift; env "$@" timeout 300 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; tail -1 $OUT/r05_gb_$tag.log; }
run_gb m6_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=150 MODE=6 SPLIT=on
run_gb m6_b2_d3_nostash SIZE=256 BATCH=2 DEPTH=3 ITERS=150 MODE=6 SPLIT=on STASH=0
run_gb m6_b1_d3 SIZE=256 BATCH=1 DEPTH=3 ITERS=150 MODE=6 SPLIT=on
run_gb m6_b2_d3_oldtowers SIZE=256 BATCH=2 DEPTH=3 ITERS=150 MODE=6 SPLIT=off
run_gb m3_eager_kernelzero SIZE=256 BATCH=1 DEPTH=1 ITERS=10 MODE=3 SPLIT=on EAGER_BETWEEN=1
run_gb m3_eager_kernelzero_nostash SIZE=256 BATCH=1 DEPTH=1 ITERS=10 MODE=3 SPLIT=on EAGER_BETWEEN=1 STASH=0
run_gb m3_eager_memset SIZE=256 BATCH=1 DEPTH=1 ITERS=10 MODE=3 SPLIT=on EAGER_BETWEEN=1 ORP_FILL=memset
run_gb m3_eager_memset_nostash SIZE=256 BATCH=1 DEPTH=1 ITERS=10 MODE=3 SPLIT=on EAGER_BETWEEN=1 STASH=0 ORP_FILL=memset
cat $OUT/r05_mfma_war.log
tail -4 $OUT/r05_soak_full_default.log; tail -4 $OUT/r05_soak_full_drain0.log
