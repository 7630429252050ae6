#!/bin/bash
# round-3 GPU call 14: Gram with XCD-grouped work items: tests + A/B on the resident 1000 x 11.2 M matrix
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "gram or aggregators or adversaries or bitwise" --timeout 120 --timeout-method thread > $O/pytest14.log 2>&1
rc=$?; tail -3 $O/pytest14.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest14.log | head -30; exit 1; fi
{
echo "== XCD-grouped work items (default)"
timeout -k 10 400 python3 tools/bench_aggregators.py 1000 resnet18 || exit 1
echo "== BLADES_GRAM_XCD=0 (dispatch order: tile pair fastest)"
BLADES_GRAM_XCD=0 timeout -k 10 400 python3 tools/bench_aggregators.py 1000 resnet18 || exit 1
} > $O/aggregators14.log 2>&1 || { tail $O/aggregators14.log; exit 1; }
grep -v amdgpu.ids $O/aggregators14.log
