#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
run_gb() { tag=$1; shift; env "$@" timeout 400 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; tail -1 $OUT/r05_gb_$tag.log | cut -c1-600; }
run_gb pp_mt3 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_DCNS_MT=3
run_gb pp_pad84 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_DCNS_PAD_LDS=84
run_gb pp_drain0 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_drain0.so
run_gb pp_default SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on
run_gb pp_mode0 SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=0 SPLIT=on
