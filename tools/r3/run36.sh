#!/bin/bash
# alternating A/B on one box: rotating activation fragments in conv_igemm_v1 (variant library) vs the shipped schedule
set -o pipefail
mkdir -p gpurun_out/r3
BLADES_HIP_LIB=tools/ab_lib/libblades_hip_afrot.so timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "conv_igemm or conv_dgrad or linear_forward" > gpurun_out/r3/pytest36.log 2>&1 || { tail -20 gpurun_out/r3/pytest36.log; exit 1; }
tail -1 gpurun_out/r3/pytest36.log
OUT=gpurun_out/r3/afrot_ab.log
: > $OUT
one() { echo -n "$1: " >> $OUT; env $2 timeout -k 10 300 python bench.py --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" >> $OUT; }
for i in 1 2 3; do one "shipped            " X=1 && one "rotating fragments " BLADES_HIP_LIB=tools/ab_lib/libblades_hip_afrot.so || exit 1; done
cat $OUT
