#!/bin/bash
# the reference's multi-local-step shapes (local20_*.yaml) and the 100 x 2 comparison point, fp32, one GPU
b() { timeout -k 10 280 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/round")'; }
echo "resnet10 100 clients x 2 steps:  $(b --model resnet --clients 100 --local-steps 2 --steps 5 --warmup 2)"
echo "resnet10  60 clients x 20 steps: $(b --model resnet --clients 60 --local-steps 20 --steps 3 --warmup 1)"
echo "cnn       60 clients x 20 steps: $(b --model cnn --dataset fashionmnist --clients 60 --local-steps 20 --steps 10 --warmup 3)"
