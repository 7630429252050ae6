#!/bin/bash
# 60-client CNN / MLP rounds under every aggregator: does the host clustering (a device sync per round) show?
set -e
mkdir -p gpurun_out; rm -f gpurun_out/run47.log
for cfg in "fashionmnist cnn" "mnist mlp"; do
  set -- $cfg
  for agg in Mean Median Trimmedmean GeoMed DnC Multikrum Centeredclipping Clippedclustering Signguard; do
    ms=$(timeout -k 10 200 python3 bench.py --dataset $1 --model $2 --clients 60 --batch 64 --steps 100 --warmup 10 --aggregator $agg 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
    echo "$2 60 clients FedSGD ALIE 20% $agg $ms ms/round" | tee -a gpurun_out/run47.log
  done
done
