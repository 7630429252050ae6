#!/bin/bash
# round-3 GPU call 16: chunk-parallel BatchNorm finalize: tests + headline A/B
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest16.log 2>&1
rc=$?; tail -3 $O/pytest16.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest16.log | head -40; exit 1; fi
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("phase_ms"))'; }
{
for i in 1 2 3; do
  echo "parallel finalize     $(b --steps 8 --warmup 2)"
  echo "serial finalize       $(BLADES_PCBN_FINALIZE=serial b --steps 8 --warmup 2)"
done
} 2>&1 | tee $O/bench_ab16.log
