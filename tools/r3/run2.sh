#!/bin/bash
# round-3 GPU call 2: new conv tests, phase profile of conv_igemm, headline A/B (r2 kernel + GradJoin + round planes vs the round-2 library)
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
R2=$(pwd)/tools/ab_lib/libblades_hip_r2.so
true
rc=0
if [ $rc -ne 0 ]; then grep -n "Error\|error\|FAILED\|assert" $O/pytest2.log | head -40; exit 1; fi
{
timeout -k 10 120 python tools/r3/igemm_prof.py --n 2048 || exit 1
timeout -k 10 120 python tools/r3/igemm_prof.py --n 12800 || exit 1
} > $O/igemm_prof.log 2>&1 || { tail -20 $O/igemm_prof.log; exit 1; }
grep -v amdgpu.ids $O/igemm_prof.log
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["phase_ms"])'; }
{
for i in 1 2 3; do
  echo "new          $(b --steps 8 --warmup 2)"
  echo "r2 lib       $(BLADES_HIP_LIB=$R2 BLADES_GRAD_JOIN=0 BLADES_ROUND_PLANES=0 b --steps 8 --warmup 2)"
done
} 2>&1 | tee $O/bench_ab2.log
