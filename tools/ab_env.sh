#!/bin/bash
# same-box sweep of one environment switch over the fp32 headline round: VAR=BLADES_X VALUES="a b c" tools/ab_env.sh
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
for rep in 1 2; do
  echo "default        $(b --steps 8 --warmup 2)"
  for v in $VALUES; do
    echo "$VAR=$v   $(env $VAR=$v timeout -k 10 200 python3 bench.py --steps 8 --warmup 2 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
  done
done
