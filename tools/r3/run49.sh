#!/bin/bash
set -e
mkdir -p gpurun_out; rm -f gpurun_out/run49.log
timeout -k 10 300 python -u -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -x -q --timeout 120 --timeout-method thread -k "small_client or dnc or aggregators_gpu or adversaries_gpu or minmax or MinMax or gram" > gpurun_out/run49_tests.log 2>&1 || { tail -40 gpurun_out/run49_tests.log; exit 1; }
tail -2 gpurun_out/run49_tests.log
for cfg in "mnist mlp" "fashionmnist cnn"; do
  set -- $cfg
  for agg in Mean GeoMed DnC Centeredclipping; do
    ms=$(timeout -k 10 200 python3 bench.py --dataset $1 --model $2 --clients 60 --batch 64 --steps 100 --warmup 10 --aggregator $agg 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')
    echo "$2 60 clients FedSGD ALIE 20% $agg $ms ms/round" | tee -a gpurun_out/run49.log
  done
done
for m in cnn; do timeout -k 10 300 python -u tools/bench_aggregators.py 60 $m 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/run49.log; done
