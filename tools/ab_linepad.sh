#!/bin/bash
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
echo "bf16 tight lines   $(b --amp bf16 --steps 8 --warmup 2)"
echo "bf16 LINEPAD=3     $(BLADES_WGRAD_LINEPAD=3 b --amp bf16 --steps 8 --warmup 2)"
echo "bf16 tight lines   $(b --amp bf16 --steps 8 --warmup 2)"
echo "fp32 tight lines   $(b --steps 8 --warmup 2)"
echo "fp32 LINEPAD=3     $(BLADES_WGRAD_LINEPAD=3 b --steps 8 --warmup 2)"
