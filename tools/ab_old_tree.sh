#!/bin/bash
# Same-box A/B of the small configurations between this tree and an older checkout under ./_old (git worktree)
b() { (cd "$1" && shift && timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'); }
small="--clients 60 --batch 64 --steps 50 --warmup 5 --aggregator Median"
for m in "mnist mlp" "fashionmnist cnn" "cifar10 cct" "cifar10 resnet"; do
  set -- $m
  echo "$2 new $(b . --dataset $1 --model $2 $small)   old $(b _old --dataset $1 --model $2 $small)   new $(b . --dataset $1 --model $2 $small)   old $(b _old --dataset $1 --model $2 $small)"
done
echo "bf16 headline new $(b . --amp bf16 --steps 8 --warmup 2)  old $(b _old --amp bf16 --steps 8 --warmup 2)"
