#!/bin/bash
# round-3 GPU call 6: register-fed weight path of conv_igemm — kernel tests, per-layer timing against the LDS-staged path
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_igemm or linear_forward or conv_epilogue or stem_as or bitwise" --timeout 120 --timeout-method thread > $O/pytest6.log 2>&1
rc=$?; tail -5 $O/pytest6.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest6.log | head -30; exit 1; fi
{
for rf in 1 0; do
  echo "== BLADES_IGEMM_RF=$rf"
  BLADES_IGEMM_RF=$rf timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --dgrad || exit 1
done
} > $O/igemm_rf6.log 2>&1 || { tail -20 $O/igemm_rf6.log; exit 1; }
grep -v amdgpu.ids $O/igemm_rf6.log
