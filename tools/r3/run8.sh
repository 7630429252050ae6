#!/bin/bash
# round-3 GPU call 8: hybrid build tests + where the register-fed path loses its time (ablations, timing only)
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_igemm or linear_forward or conv_epilogue or stem_as or bitwise" --timeout 120 --timeout-method thread > $O/pytest8.log 2>&1
rc=$?; tail -3 $O/pytest8.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest8.log | head -30; exit 1; fi
BLADES_IGEMM_RF=1 timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_igemm or linear_forward or conv_epilogue or stem_as" --timeout 120 --timeout-method thread > $O/pytest8b.log 2>&1
rc=$?; tail -3 $O/pytest8b.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest8b.log | head -30; exit 1; fi
{
echo "== hybrid (default)"
timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --dgrad --layers 0 1 2 3 4 5 6 8 || exit 1
for ab in 0 1 8 32; do
  echo "== BLADES_IGEMM_RF=1 BLADES_IGEMM_ABLATE=$ab (1: no staging at all, 8: no weight loads, 32: no input restage)"
  BLADES_IGEMM_RF=1 BLADES_IGEMM_ABLATE=$ab timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --layers 0 1 2 3 || exit 1
done
for ab in 0 1 8 32; do
  echo "== BLADES_IGEMM_RF=0 BLADES_IGEMM_ABLATE=$ab"
  BLADES_IGEMM_RF=0 BLADES_IGEMM_ABLATE=$ab timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --layers 0 1 2 3 || exit 1
done
} > $O/igemm_ab8.log 2>&1 || { tail -20 $O/igemm_ab8.log; exit 1; }
grep -v amdgpu.ids $O/igemm_ab8.log
