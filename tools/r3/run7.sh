#!/bin/bash
# round-3 GPU call 7: same-box A/B of conv_igemm builds (per-layer timing, interleaved twice)
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
L=$(pwd)/tools/ab_lib
{
for rep in 1 2; do
for v in r2 rf_noahead cur; do
  lib=$L/libblades_hip_$v.so; [ $v = cur ] && lib=""
  echo "== $v (rep $rep)"
  BLADES_HIP_LIB=$lib timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --dgrad --layers 0 1 2 3 4 5 6 8 || exit 1
done
done
} > $O/igemm_ab7.log 2>&1 || { tail -20 $O/igemm_ab7.log; exit 1; }
grep -v amdgpu.ids $O/igemm_ab7.log
