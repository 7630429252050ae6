#!/bin/bash
set -e
mkdir -p gpurun_out; rm -f gpurun_out/run55.log
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 180 --timeout-method thread -k "row_ or rowstats or sqdist or aggregators_gpu or adversaries_gpu or bitwise_deterministic" > gpurun_out/run55_tests.log 2>&1 || { tail -40 gpurun_out/run55_tests.log; exit 1; }
tail -2 gpurun_out/run55_tests.log
timeout -k 10 200 python -u tools/r3/gram_small_probe.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/run55.log
timeout -k 10 500 python -u tools/bench_aggregators.py 1000 resnet18 2>&1 | grep -v amdgpu | tee -a gpurun_out/run55.log
