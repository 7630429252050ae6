#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "dnc or small_client or aggregators_gpu or DnC" > gpurun_out/run40_tests.log 2>&1 || { tail -40 gpurun_out/run40_tests.log; exit 1; }
tail -3 gpurun_out/run40_tests.log
for m in mlp cnn resnet; do timeout -k 10 300 python -u tools/bench_aggregators.py 60 $m 2>&1 | grep -i "K=\|DnC" | tee -a gpurun_out/run40_agg.log; done
timeout -k 10 300 python -u tools/bench_aggregators.py 1000 resnet18 2>&1 | grep -i "K=\|DnC" | tee -a gpurun_out/run40_agg.log
