#!/bin/bash
# Full GPU suite three ways (run on the GPU box through gpurun; logs -> gpurun_out/, copied to profiles/r0N_gputest_*.log):
#   forward   collection order, no -x (one failure never hides the rest)
#   reverse   ORP_TEST_ORDER=reverse (tests/conftest.py): no test may depend on allocator / cache state of an earlier one
#   perfile   every test file alone in a fresh process
# usage: tools/gpu_suite.sh <tag>     e.g. r03
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
python -c "import orientedreppoints_amd._lib as L; print('orp_version', L.lib().orp_version().decode())" > $OUT/${TAG}_gputest_forward.log 2>&1
python -m pytest tests -m gpu -q -p no:cacheprovider -rfE 2>&1 | tail -60 >> $OUT/${TAG}_gputest_forward.log
ORP_TEST_ORDER=reverse python -m pytest tests -m gpu -q -p no:cacheprovider -rfE 2>&1 | tail -60 > $OUT/${TAG}_gputest_reverse.log
: > $OUT/${TAG}_gputest_perfile.log
for f in tests/test_*.py; do
  if grep -q "mark.gpu" $f; then
    echo "== $f" >> $OUT/${TAG}_gputest_perfile.log
    python -m pytest $f -m gpu -q -p no:cacheprovider -rfE 2>&1 | tail -4 >> $OUT/${TAG}_gputest_perfile.log
  fi
done
tail -3 $OUT/${TAG}_gputest_forward.log; tail -3 $OUT/${TAG}_gputest_reverse.log; grep -E "==|passed|failed" $OUT/${TAG}_gputest_perfile.log
