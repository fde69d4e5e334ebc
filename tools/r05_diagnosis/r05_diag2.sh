#!/bin/bash
# Round-5 diagnosis pass 2 (GPU box): minimal memset-node repro; coefficient-table / A-tile-row traces of failing MT = 1 DeformConv
# split launches (-DORP_DCNS_DRAIN=0 builds); graph replays vs eager with the tile height / residency forced.
OUT=gpurun_out
mkdir -p $OUT
(cd tests/checks && timeout 120 ./graph_memset_node) > $OUT/r05_graph_memset_node.log 2>&1
run_tr() { tag=$1; shift; env "$@" timeout 300 python tests/checks/split_trace.py > $OUT/r05_trace_$tag.log 2>&1; }
run_tr d0t2_p6_b1 ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace2.so NPROD=6 BATCH=1 N=3000
run_tr d0t2_p6_b2 ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace2.so NPROD=6 BATCH=2 N=3000
run_tr d0t1_p6_b2 ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace1.so NPROD=6 BATCH=2 N=3000
run_tr d0t2_p3_b1 ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace2.so NPROD=3 BATCH=1 N=300
run_tr d0t1_p3_b1 ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace1.so NPROD=3 BATCH=1 N=300
run_gb() { tag=$1; shift; env "$@" timeout 300 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; tail -1 $OUT/r05_gb_$tag.log | cut -c1-400; }
run_gb m6_b2_d3_mt3 SIZE=256 BATCH=2 DEPTH=3 ITERS=150 MODE=6 SPLIT=on ORP_DCNS_MT=3
run_gb m6_b2_d3_pad84 SIZE=256 BATCH=2 DEPTH=3 ITERS=150 MODE=6 SPLIT=on ORP_DCNS_PAD_LDS=84
run_gb m6_b2_d1 SIZE=256 BATCH=2 DEPTH=1 ITERS=300 MODE=6 SPLIT=on
run_gb m0_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=150 MODE=0 SPLIT=on
cat $OUT/r05_graph_memset_node.log | head -60
for f in $OUT/r05_trace_*.log; do echo "== $f"; cut -c1-1500 $f | head -40; done
