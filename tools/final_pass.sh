#!/bin/bash
# The round's measurement set at HEAD (run on the GPU box through gpurun; everything lands in gpurun_out/, the summaries are
# then folded into profiles/<tag>_* locally -- see the tail of this file).  usage: tools/final_pass.sh <tag>
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=gpurun_out
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${TAG}_smoke.log 2>&1
# counter passes first: bench.py quotes `traffic` from profiles/<tag>_pmc.json only when it was collected on the running build
bash tools/pmc_all.sh > $O/${TAG}_pmc_all.log 2>&1
python tools/make_pmc_json.py profiles/${TAG} $O/pmc_fetch $O/pmc_write $O/pmc_sqa $O/pmc_sqb > /dev/null 2>&1
cp profiles/${TAG}_pmc.json profiles/${TAG}_pmc.txt $O/ 2>/dev/null
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --model r101 --batch 2 --steps 60 --no-cpu-baseline > $O/${TAG}_bench_r101_b2.json 2>/dev/null
python bench.py --size 1536 --steps 60 --no-cpu-baseline > $O/${TAG}_bench_1536.json 2>/dev/null
python bench.py --mode train --steps 30 > $O/${TAG}_bench_train.json 2>/dev/null
python bench.py --mode train --dtype fp16 --steps 30 > $O/${TAG}_bench_train_fp16.json 2>/dev/null
python bench.py --mode train --dtype bf16 --steps 30 > $O/${TAG}_bench_train_bf16.json 2>/dev/null
python bench.py --model swin_t --mode train --dtype fp16 --size 1024,1536 --steps 20 --warmup 4 > $O/${TAG}_bench_swin_train.json 2>/dev/null
python bench.py --model r50_dcnv2 --mode train --steps 12 --warmup 3 > $O/${TAG}_bench_r50dcnv2_train.json 2>/dev/null
python bench.py --gpus 2 --device cpu --dry --mode train 2>/dev/null | grep "^{" > $O/${TAG}_bench_dry_2ranks_train.json
# every result compared, >= 10 000 launches per configuration (tile heights 1 and 3, one and two images, 3 and 6 products, the
# DeformConv and the convolution instantiation), next to a GEMM stream and a second stream of the same kernel
SOAK_N=10000 timeout 900 python tests/checks/soak_split_full.py > $O/${TAG}_soak.log 2>&1
SOAK_N=1000 timeout 600 python tests/checks/soak_dcn_wgrad16.py > $O/${TAG}_soak_dcn_wgrad16.log 2>&1
(cd tests/checks && timeout 200 ./mfma_refill_victim 100) > $O/${TAG}_mfma_refill_victim.log 2>&1
AGGR=none,gemm,conv_big,conv_small,dcn_small N=300 timeout 200 python tests/checks/victim_probe.py > $O/${TAG}_victim_probe.log 2>&1
SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=3 SPLIT=auto timeout 300 python tests/checks/graph_bitwise.py > $O/${TAG}_graph_bitwise_mode3.log 2>&1
SIZE=256 BATCH=2 DEPTH=3 ITERS=2000 MODE=6 SPLIT=auto timeout 300 python tests/checks/graph_bitwise.py > $O/${TAG}_graph_bitwise_mode6.log 2>&1
SIZE=1024 BATCH=1 DEPTH=4 ITERS=150 NIMG=4 MODE=3 SPLIT=auto timeout 300 python tests/checks/graph_bitwise.py > $O/${TAG}_graph_bitwise_1024.log 2>&1
python tools/time_convex.py > $O/${TAG}_convex.log 2>&1
bash tests/checks/clock_under_split.sh > $O/${TAG}_clock_under_split.log 2>&1
python tests/checks/time_dcn_backward.py > $O/${TAG}_dcn_backward.log 2>&1
python tests/checks/time_dcn_pair.py > $O/${TAG}_dcn_pair.log 2>&1
ORP_DCN_SPLIT=0 python tests/checks/time_dcn_pair.py >> $O/${TAG}_dcn_pair.log 2>&1
ORP_DCN_SPLIT=3 python tests/checks/time_dcn_pair.py >> $O/${TAG}_dcn_pair.log 2>&1
ORP_DCN_SPLIT=6 python tests/checks/time_dcn_pair.py >> $O/${TAG}_dcn_pair.log 2>&1
ORP_DCN_SPLIT=9 python tests/checks/time_dcn_pair.py >> $O/${TAG}_dcn_pair.log 2>&1
python tests/checks/time_towers.py > $O/${TAG}_towers.log 2>&1
(python tests/checks/time_ws.py; ORP_DCNS_WS=1 python tests/checks/time_ws.py) 2>&1 | grep '^\[' > $O/${TAG}_ws_vs_symmetric.log
python tests/checks/time_minarearect.py 2>&1 | grep sets > $O/${TAG}_minarearect.log
(hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_repro docs/mi355x_pk_mul_f32_next_to_mfma.hip 2>/dev/null && timeout 200 /tmp/pk_repro 100) > $O/${TAG}_pk_mul_reproducer.log 2>&1
python tests/checks/time_wgrad.py > $O/${TAG}_wgrad.log 2>&1
ORP_DCN_SPLIT=6 python bench.py --steps 100 --no-cpu-baseline > $O/${TAG}_bench_mode6.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${TAG}_prof_bench -- python $R/bench.py --steps 30 --no-cpu-baseline --pipeline 1 > $R/$O/${TAG}_prof_bench.log 2>&1)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/${TAG}_prof_train -- python $R/bench.py --mode train --steps 12 > $R/$O/${TAG}_prof_train.log 2>&1)
tail -2 $O/${TAG}_smoke.log; tail -c 600 $O/${TAG}_bench.json; echo; tail -c 300 $O/${TAG}_bench_train.json
# locally afterwards (profiles/ on the GPU box is a scratch copy):
#   python tools/make_pmc_json.py profiles/<tag> gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sqa gpurun_out/pmc_sqb
#   python tools/summarize_prof.py gpurun_out/<tag>_prof_bench profiles/<tag>_bench
#   python tools/summarize_train_prof.py gpurun_out/<tag>_prof_train profiles/<tag>_train_step.txt
