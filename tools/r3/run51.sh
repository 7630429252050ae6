#!/bin/bash
set -e
mkdir -p gpurun_out; rm -f gpurun_out/run51.log
timeout -k 10 600 python -u -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -x -q --timeout 180 --timeout-method thread -k "round_plane or nonfinite or optimizer_steps or graph_replay or graphed_local or auto_mode" > gpurun_out/run51_tests.log 2>&1 || { tail -40 gpurun_out/run51_tests.log; exit 1; }
tail -2 gpurun_out/run51_tests.log
for cfg in "mnist mlp" "fashionmnist cnn" "cifar10 cct" "cifar10 resnet"; do
  set -- $cfg
  for agg in Mean Median DnC; do
    ms=$(timeout -k 10 200 python3 bench.py --dataset $1 --model $2 --clients 60 --batch 64 --steps 100 --warmup 10 --aggregator $agg 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
    echo "$2 60 clients FedSGD ALIE 20% $agg $ms ms/round" | tee -a gpurun_out/run51.log
  done
done
