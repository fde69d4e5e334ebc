#!/bin/bash
OUT=gpurun_out
mkdir -p $OUT
b() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --train-probe 0 2>/dev/null | tail -1 > $OUT/r05_bench_$tag.json; python - $OUT/r05_bench_$tag.json $tag <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], 'value', d.get('value'), 'serial', d.get('value_serial'), 'ms', d.get('ms_per_step'), 'graph_ms', d.get('graph_replay_ms'), 'roof', d.get('roofline',{}).get('frac'), d.get('config',{}).get('arithmetic_mode'))
PY
}
b drain2
b drain0 ORP_HIP_LIB=build_variants/liborp_hip_drain0.so
b fence ORP_HIP_LIB=build_variants/liborp_hip_fence.so
b noown_drain0 ORP_HIP_LIB=build_variants/liborp_hip_noown_drain0.so
b mode3_drain2 ORP_DCN_SPLIT=3
b mode3_drain0 ORP_DCN_SPLIT=3 ORP_HIP_LIB=build_variants/liborp_hip_drain0.so
SIZE=256 BATCH=1 DEPTH=1 ITERS=20 MODE=3 SPLIT=on EAGER_BETWEEN=1 ORP_FILL=memset timeout 200 python tests/checks/graph_bitwise.py 2>&1 | tail -1 | cut -c1-300
