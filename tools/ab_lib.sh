#!/bin/bash
# same-box A/B of two builds of libblades_hip.so: tools/ab_lib/libblades_hip_prev.so against the in-tree one
#   AMP=fp32|bf16 tools/ab_lib.sh
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
PREV=$(pwd)/tools/ab_lib/libblades_hip_prev.so
AMP=${AMP:-fp32}
for i in 1 2 3; do
  echo "$AMP new   $(b --amp $AMP --steps 8 --warmup 2)"
  echo "$AMP prev  $(BLADES_HIP_LIB=$PREV b --amp $AMP --steps 8 --warmup 2)"
done
