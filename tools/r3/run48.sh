#!/bin/bash
# kernel table of the fp32 headline round, final build, default chunking, ONE chunk stream (un-overlapped durations)
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
BLADES_CHUNK_STREAMS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof48 -o b -- python3 $R/bench.py --steps 3 --warmup 1 > $R/gpurun_out/run48_bench.log 2>&1
grep '"metric"' $R/gpurun_out/run48_bench.log | cut -c1-200
f=$(find /tmp/prof48 -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $R/gpurun_out/run48_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 4 rounds (1 warm-up + 3 timed): {tot/4e6:.1f} ms per round")
for r in rows[:30]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f} %  {float(r['TotalDurationNs'])/1e6/4:8.2f} ms/round  {int(r['Calls'])//4:5d} calls/round  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:120]}")
PY
cat $R/gpurun_out/run48_kernels.txt | cut -c1-200
