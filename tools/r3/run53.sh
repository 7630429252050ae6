#!/bin/bash
set -e
mkdir -p gpurun_out; rm -f gpurun_out/run53.log
timeout -k 10 600 python -u -m pytest tests/test_engine_gpu.py -x -q --timeout 180 --timeout-method thread -k "deferred or round_plane or nonfinite" > gpurun_out/run53_tests.log 2>&1 || { tail -40 gpurun_out/run53_tests.log; exit 1; }
tail -2 gpurun_out/run53_tests.log
for lag in 0 1; do
for cfg in "mnist mlp" "fashionmnist cnn" "cifar10 cct" "cifar10 resnet"; do
  set -- $cfg
  ms=$(BLADES_METRICS_LAG=$lag timeout -k 10 200 python3 bench.py --dataset $1 --model $2 --clients 60 --batch 64 --steps 100 --warmup 10 --aggregator Median 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
  echo "BLADES_METRICS_LAG=$lag $2 60 clients FedSGD ALIE 20% Median $ms ms/round" | tee -a gpurun_out/run53.log
done
done
