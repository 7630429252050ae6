#!/bin/bash
# fp32 headline round over (GPU_MAX_HW_QUEUES : chunk streams : clients per chunk) on one box
for cfg in ${SWEEP:-4:3:32 8:3:32 8:4:32 8:5:32 8:6:32 8:4:25 8:6:20 4:3:32}; do
  IFS=: read q s c <<< "$cfg"
  line=$(GPU_MAX_HW_QUEUES=$q BLADES_CHUNK_STREAMS=$s timeout -k 10 200 python -u bench.py --steps 6 --warmup 2 --chunk $c 2>/dev/null | tail -1) || exit 1
  echo "queues $q streams $s chunk $c: $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms  peak", d.get("peak_hbm_gb"), "GB")')"
done
