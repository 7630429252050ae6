#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/ -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/run39_tests.log 2>&1 || { tail -30 gpurun_out/run39_tests.log; exit 1; }
tail -3 gpurun_out/run39_tests.log
echo "== small kernel" && timeout -k 10 200 python -u tools/r3/gram_small_probe.py 2>&1 | tee gpurun_out/run39_small.log
for m in cnn resnet; do timeout -k 10 300 python -u tools/bench_aggregators.py 60 $m 2>&1 | tee gpurun_out/run39_agg_$m.log; done
