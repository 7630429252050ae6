#!/bin/bash
# A/B on one box: shortcut branch built after conv1 (BatchNorm-backward sums of the block input from the 3x3 data-gradient kernel)
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q --timeout 200 --timeout-method thread -k "batchnorm or resnet or fp64 or bitwise or graph or stream" > gpurun_out/r3/pytest34.log 2>&1 || { tail -30 gpurun_out/r3/pytest34.log; exit 1; }
tail -2 gpurun_out/r3/pytest34.log
OUT=gpurun_out/r3/sc_order_ab.log
: > $OUT
for i in 1 2; do
  for v in new; do
    echo -n "$v: " >> $OUT
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" >> $OUT || exit 1
  done
  echo -n "BN sums in dgrad off: " >> $OUT
  BLADES_BN_BWD_IN_DGRAD=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" >> $OUT || exit 1
done
cat $OUT
