#!/bin/bash
# Power / clock samples of the GPU while the headline bench runs (is the 2.0 GHz shader clock a power limit?)
set -o pipefail
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/power_samples.log
: > $OUT
rocm-smi --showmaxpower --showpower --showclocks > gpurun_out/r3/power_idle.log 2>&1
( while true; do
    echo "t=$(date +%s.%N)" >> $OUT
    rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" >> $OUT
    sleep 0.4
  done ) &
SAMPLER=$!
timeout -k 10 300 python bench.py --steps 12 --warmup 2 > gpurun_out/r3/power_bench.json 2> gpurun_out/r3/power_bench.err
rc=$?
kill $SAMPLER
wait $SAMPLER 2>/dev/null
tail -1 gpurun_out/r3/power_bench.json | cut -c1-200
exit $rc
