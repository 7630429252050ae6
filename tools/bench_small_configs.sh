#!/bin/bash
# Rounds/s of the reference's 60-client experiment shapes (tuned_examples/*.yaml) on one GPU.
set -e
out=${1:-gpurun_out/small_configs.log}
: > "$out"
for cfg in "mnist mlp" "fashionmnist cnn" "cifar10 resnet" "cifar10 cct"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --dataset "$1" --model "$2" --clients 60 --batch 64 --steps 50 --warmup 5 \
      --aggregator Median 2>>"$out.err" | tee -a "$out" | cut -c1-160
done
