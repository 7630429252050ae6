#!/bin/bash
# bf16 secondary mode: kernel table of the headline config
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof42 -o b -- python3 $R/bench.py --steps 3 --warmup 1 --amp bf16 > $R/gpurun_out/run42_bench.log 2>&1
tail -2 $R/gpurun_out/run42_bench.log | cut -c1-300
f=$(find /tmp/prof42 -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/run42_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 4 rounds (1 warm-up + 3)")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f} %  {float(r['TotalDurationNs'])/1e6/4:8.2f} ms/round  {int(r['Calls'])//4:5d} calls/round  {r['Name'][:110]}")
PY
cat $R/gpurun_out/run42_kernels.txt
