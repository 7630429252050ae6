#!/bin/bash
# round-3 GPU call 12: full GPU suite with the fp64-anchored engine tests
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -q -s --timeout 300 --timeout-method thread > $O/pytest12.log 2>&1
rc=$?; tail -5 $O/pytest12.log; grep -n "fp64 anchor" $O/pytest12.log | cut -c1-700
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest12.log | head -40; exit 1; fi
