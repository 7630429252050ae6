#!/bin/bash
# round-3 GPU call 11: prologue with all first-chunk loads in flight before the LDS clear: tests + same-box per-layer A/B
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_igemm or linear_forward or conv_epilogue or conv_dgrad or stem_as or bitwise" --timeout 120 --timeout-method thread > $O/pytest11.log 2>&1
rc=$?; tail -3 $O/pytest11.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest11.log | head -30; exit 1; fi
L=$(pwd)/tools/ab_lib
{
for rep in 1 2; do
for v in prev cur; do
  lib=$L/libblades_hip_$v.so; [ $v = cur ] && lib=""
  echo "== $v (rep $rep)"
  BLADES_HIP_LIB=$lib timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --dgrad --layers 0 1 2 3 4 5 6 8 || exit 1
done
done
} > $O/igemm_ab11.log 2>&1 || { tail -20 $O/igemm_ab11.log; exit 1; }
