#!/bin/bash
# Round-5 diagnosis pass 4 (GPU box): VALU-write-behind-MFMA probe; graph replays after the memset-node removal; the GPU suite.
OUT=gpurun_out
mkdir -p $OUT
(cd tests/checks && timeout 300 ./mfma_war 40 2000) > $OUT/r05_mfma_war2.log 2>&1
grep -A8 "part C" $OUT/r05_mfma_war2.log
run_gb() { tag=$1; shift; env "$@" timeout 300 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; tail -1 $OUT/r05_gb_$tag.log | cut -c1-400; }
run_gb fill_m6_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=600 MODE=6 SPLIT=on
run_gb fill_m6_b2_d3_memset SIZE=256 BATCH=2 DEPTH=3 ITERS=600 MODE=6 SPLIT=on ORP_FILL=memset
run_gb fill_m3_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=300 MODE=3 SPLIT=on
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -15 > $OUT/r05_gputest_a.log; tail -5 $OUT/r05_gputest_a.log
