#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "conv_igemm or stem_as" > gpurun_out/run43_tests.log 2>&1 || { tail -40 gpurun_out/run43_tests.log; exit 1; }
tail -3 gpurun_out/run43_tests.log
timeout -k 10 300 python -u bench.py --steps 5 --warmup 2 --amp bf16 2>&1 | tail -1 | cut -c1-200
BLADES_IGEMM_CR=0 timeout -k 10 300 python -u bench.py --steps 5 --warmup 2 --amp bf16 2>&1 | tail -1 | cut -c1-200
