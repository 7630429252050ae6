#!/bin/bash
# round-3 GPU call 9: headline round, same-box A/B: hybrid conv_igemm (default) vs all-RF vs the round-2 kernel library
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
R2=$(pwd)/tools/ab_lib/libblades_hip_r2.so
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("phase_ms"))'; }
{
for i in 1 2 3; do
  echo "hybrid       $(b --steps 8 --warmup 2)"
  echo "all RF       $(BLADES_IGEMM_RF=1 b --steps 8 --warmup 2)"
  echo "r2 igemm     $(BLADES_HIP_LIB=$R2 b --steps 8 --warmup 2)"
done
} 2>&1 | tee $O/bench_ab9.log
