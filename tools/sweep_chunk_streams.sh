#!/bin/bash
# fp32 headline round over (clients per chunk : chunk streams) on one box:
#   SWEEP="40:2 25:3" tools/sweep_chunk_streams.sh > gpurun_out/sweep.txt
for cfg in ${SWEEP:-40:2 40:3 32:2 50:2 25:2 25:3 40:2}; do
  c=${cfg%%:*}; s=${cfg##*:}
  line=$(BLADES_CHUNK_STREAMS=$s timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --chunk $c 2>/dev/null | tail -1) || exit 1
  echo "chunk $c streams $s: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  peak", d.get("peak_hbm_gb"), "GB")')"
done
