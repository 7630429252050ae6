#!/bin/bash
# round-3 GPU call 1: full GPU suite, conv_igemm microbench (persistent vs one-tile-per-workgroup vs the round-2 library), headline A/B
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
R2=$(pwd)/tools/ab_lib/libblades_hip_r2.so
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest1.log 2>&1
rc=$?; tail -5 $O/pytest1.log
if [ $rc -ne 0 ]; then grep -n "Error\|error\|FAILED\|assert" $O/pytest1.log | head -40; exit 1; fi
{
for n in 2048 12800; do
  echo "== persistent (new), n=$n";   timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n $n --layers 0 1 2 3 4 5 6 8 --dgrad || exit 1
  echo "== one tile per workgroup (BLADES_IGEMM_GRID=0), n=$n"; BLADES_IGEMM_GRID=0 timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n $n --layers 0 1 2 3 4 5 6 8 --dgrad || exit 1
  echo "== round-2 library, n=$n";    BLADES_HIP_LIB=$R2 timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n $n --layers 0 1 2 3 4 5 6 8 --dgrad || exit 1
done
} > $O/igemm_time1.log 2>&1 || { tail -20 $O/igemm_time1.log; exit 1; }
cat $O/igemm_time1.log
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["phase_ms"])'; }
{
for i in 1 2 3; do
  echo "new          $(b --steps 8 --warmup 2)"
  echo "new grid0    $(BLADES_IGEMM_GRID=0 b --steps 8 --warmup 2)"
  echo "new nojoin   $(BLADES_GRAD_JOIN=0 BLADES_ROUND_PLANES=0 b --steps 8 --warmup 2)"
  echo "r2 lib       $(BLADES_HIP_LIB=$R2 BLADES_GRAD_JOIN=0 BLADES_ROUND_PLANES=0 b --steps 8 --warmup 2)"
done
} 2>&1 | tee $O/bench_ab1.log
