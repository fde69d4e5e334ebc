#!/bin/bash
# Round-5 diagnosis pass 3 (GPU box): the mask sequence in isolation; the select-free coefficient table under soak (with and without
# the round-4 "drain", next to the old formulation as the control); mode-3 trace against a six-product reference table; graph replays.
OUT=gpurun_out
mkdir -p $OUT
(cd tests/checks && timeout 300 ./sgpr_mask_probe 20 20000) > $OUT/r05_sgpr_mask_probe.log 2>&1
run_soak() { tag=$1; shift; env "$@" KINDS=dcn SOAK_N=3000 timeout 400 python tests/checks/soak_split_full.py > $OUT/r05_soak_$tag.log 2>&1; tail -1 $OUT/r05_soak_$tag.log; }
run_soak fix_drain2
run_soak fix_drain0 ORP_HIP_LIB=build_variants/liborp_hip_drain0.so
run_soak sel_drain0 ORP_HIP_LIB=build_variants/liborp_hip_drain0_sel.so
run_soak sel_drain2 ORP_HIP_LIB=build_variants/liborp_hip_sel.so
run_tr() { tag=$1; shift; env "$@" timeout 300 python tests/checks/split_trace.py > $OUT/r05_trace_$tag.log 2>&1; }
run_tr sel_d0_p3_b1 ORP_HIP_LIB=build_variants/liborp_hip_drain0_sel_trace2.so NPROD=3 BATCH=1 N=300
run_tr fix_d0_p3_b1 ORP_HIP_LIB=build_variants/liborp_hip_drain0_trace2.so NPROD=3 BATCH=1 N=300
run_gb() { tag=$1; shift; env "$@" timeout 300 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; tail -1 $OUT/r05_gb_$tag.log | cut -c1-400; }
run_gb fix_m6_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=300 MODE=6 SPLIT=on
run_gb fix_m6_b1_d3 SIZE=256 BATCH=1 DEPTH=3 ITERS=300 MODE=6 SPLIT=on
run_gb sel_m6_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=300 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_sel.so
run_gb fix_m3_b2_d3 SIZE=256 BATCH=2 DEPTH=3 ITERS=300 MODE=3 SPLIT=on
run_gb fix_m3_b1_d1_eager SIZE=256 BATCH=1 DEPTH=1 ITERS=20 MODE=3 SPLIT=on EAGER_BETWEEN=1
cat $OUT/r05_sgpr_mask_probe.log
for f in $OUT/r05_trace_sel_d0_p3_b1.log $OUT/r05_trace_fix_d0_p3_b1.log; do echo "== $f"; cut -c1-700 $f | head -24; done
grep -h "differs from the first one: [1-9]" $OUT/r05_soak_*.log | cut -c1-300 | head -20
