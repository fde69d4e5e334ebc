#!/bin/bash
# the library without packed fp32 instructions: every former failure scenario, with every other protection switched OFF ("worst":
# no SIMD ownership, no fence, the select formulation of the coefficient table), next to the same build WITH packed instructions
OUT=gpurun_out
mkdir -p $OUT
run_soak() { tag=$1; shift; env "$@" KINDS=dcn SOAK_N=2000 timeout 400 python tests/checks/soak_split_full.py > $OUT/r05_soak_$tag.log 2>&1; echo "soak $tag: $(tail -1 $OUT/r05_soak_$tag.log)"; }
run_soak nopk_worst ORP_HIP_LIB=build_variants/liborp_hip_worst.so
run_soak pk_worst ORP_HIP_LIB=build_variants/liborp_hip_worst_pk.so
run_gb() { tag=$1; shift; env "$@" timeout 300 python tests/checks/graph_bitwise.py > $OUT/r05_gb_$tag.log 2>&1; echo "graph $tag: $(tail -1 $OUT/r05_gb_$tag.log | cut -c90-500)"; }
run_gb nopk_worst SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_worst.so
run_gb pk_worst SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_worst_pk.so
run_gb nopk_worst_m3 SIZE=256 BATCH=2 DEPTH=3 ITERS=1000 MODE=3 SPLIT=on ORP_HIP_LIB=build_variants/liborp_hip_worst.so
run_gb nopk_intree SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=on
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -4
