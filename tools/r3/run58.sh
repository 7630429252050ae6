#!/bin/bash
# idle gaps of the shipping fp32 headline round (three chunk streams), final build
set -e
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof58 -o t -- python3 $R/bench.py --steps 4 --warmup 1 > $R/gpurun_out/run58_bench.log 2>&1
grep '"metric"' $R/gpurun_out/run58_bench.log | cut -c1-160
f=$(find /tmp/prof58 -name "*kernel_trace.csv" | head -1)
python3 $R/tools/r3/gap_stats.py "$f" 0.5 12 > $R/gpurun_out/run58_gaps.txt
head -30 $R/gpurun_out/run58_gaps.txt | cut -c1-180
