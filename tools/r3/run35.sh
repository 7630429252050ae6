#!/bin/bash
# alternating A/B on one box: shortcut branch built after conv1 (new) vs before (BLADES_SHORTCUT_FIRST=1)
set -o pipefail
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/sc_order_ab.log
: > $OUT
one() { echo -n "$1: " >> $OUT; env $2 timeout -k 10 300 python bench.py --steps 12 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'])" >> $OUT; }
for i in 1 2 3; do one "conv1 first (new)     " X=1 && one "shortcut first (old)  " BLADES_SHORTCUT_FIRST=1 || exit 1; done
cat $OUT
