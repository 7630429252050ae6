#!/bin/bash
# round-3 GPU call 17: conflict-free pixel -> fragment assignment: tests, per-layer A/B, PMC conflict counters
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_igemm or linear_forward or conv_epilogue or conv_dgrad or stem_as or bitwise" --timeout 120 --timeout-method thread > $O/pytest17.log 2>&1
rc=$?; tail -3 $O/pytest17.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest17.log | head -30; exit 1; fi
{
for rep in 1 2; do
for v in 1 0; do
  echo "== BLADES_IGEMM_PERM=$v (rep $rep)"
  BLADES_IGEMM_PERM=$v timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --dgrad --layers 2 3 5 6 || exit 1
done
done
} > $O/igemm_ab17.log 2>&1 || { tail -20 $O/igemm_ab17.log; exit 1; }
grep -v amdgpu.ids $O/igemm_ab17.log
bash tools/r3/profile_pmc3.sh > /dev/null || exit 1
grep "igemm" $O/pmc/table.txt
