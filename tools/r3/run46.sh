#!/bin/bash
# regression nets with the final build: YAML grid sample + shape fuzz
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -u tools/yaml_sweep_smoke.py --per-file 6 --rounds 2 > gpurun_out/run46_yaml.log 2>&1; rc1=$?
tail -3 gpurun_out/run46_yaml.log | cut -c1-200
[ $rc1 -ne 0 ] && { grep -n "FAIL\|Error\|Traceback" gpurun_out/run46_yaml.log | head -20; exit 1; }
timeout -k 10 400 python -u tools/shape_fuzz_smoke.py > gpurun_out/run46_fuzz.log 2>&1; rc2=$?
tail -3 gpurun_out/run46_fuzz.log | cut -c1-200
[ $rc2 -ne 0 ] && { grep -n "FAIL\|Error\|Traceback" gpurun_out/run46_fuzz.log | head -20; exit 1; }
exit 0
