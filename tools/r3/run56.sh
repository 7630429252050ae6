#!/bin/bash
# kernel tables of the general path (local steps > 1) with the final build
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/run56_tables.txt
for cfg in "fashionmnist cnn 20 10" "cifar10 resnet 20 4" "cifar10 cct 4 6"; do
  set -- $cfg
  rm -rf /tmp/prof56
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof56 -o b -- python3 $R/bench.py --dataset $1 --model $2 --clients 60 --batch 64 --local-steps $3 --steps $4 --warmup 4 --aggregator Median > $R/gpurun_out/run56_bench_$2.log 2>&1
  f=$(find /tmp/prof56 -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$2 60 x $3" $(( $4 + 4 )) >> $R/gpurun_out/run56_tables.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[3])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"== {sys.argv[2]} local steps: kernel time {tot/1e6/n:.2f} ms per round ({n} rounds incl. warm-up)")
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs'])/tot*100:5.1f} %  {float(r['TotalDurationNs'])/1e6/n:8.3f} ms/round  {int(r['Calls'])//n:5d} calls/round  avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
  grep '"metric"' $R/gpurun_out/run56_bench_$2.log | cut -c1-150 >> $R/gpurun_out/run56_tables.txt
done
cat $R/gpurun_out/run56_tables.txt | cut -c1-200
