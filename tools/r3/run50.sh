#!/bin/bash
# timeline of one MLP round with DnC: where do the 0.6 ms over Mean go?
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof50 -o t -- python3 $R/bench.py --dataset mnist --model mlp --clients 60 --batch 64 --steps 20 --warmup 10 --aggregator Mean > $R/gpurun_out/run50_bench.log 2>&1
f=$(find /tmp/prof50 -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/run50_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last round: find the last occurrence of the first kernel of a round (augment or sampler gather); print the last ~120 kernels
tail = rows[-100:]
t0 = int(tail[0]['Start_Timestamp'])
prev_end = t0
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0)/1e3:9.1f} us  dur {(e - s)/1e3:7.1f}  gap {(s - prev_end)/1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev_end = max(prev_end, e)
PY
tail -100 $R/gpurun_out/run50_timeline.txt | cut -c1-150
