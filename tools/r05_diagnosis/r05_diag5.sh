#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
run_gb() { tag=$1; shift; env "$@" timeout 400 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; tail -1 $OUT/r05_gb_$tag.log | cut -c1-600; }
run_gb pp_m6_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=1500 MODE=6 SPLIT=on
run_gb pp_m6_b2_d3_oldtowers SIZE=256 BATCH=2 DEPTH=3 ITERS=1000 MODE=6 SPLIT=off
grep -B1 -A12 "^iteration" $OUT/r05_gb_pp_m6_b2_d3.log | cut -c1-900 | head -120
