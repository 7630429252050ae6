#!/bin/bash
# round-3 GPU call 19: vectorised weight-plane prep: tests + general-path timings (prev library vs current)
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_ops_gpu.py tests/test_engine_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest19.log 2>&1
rc=$?; tail -3 $O/pytest19.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest19.log | head -40; exit 1; fi
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
{
for i in 1 2; do
echo "cnn 60x20 local steps        $(b --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 2 --aggregator Median)"
echo "cnn 60x20 library linear     $(BLADES_LINEAR_IMPL=lib b --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 2 --aggregator Median)"
echo "mlp 60x20 local steps        $(b --dataset mnist --model mlp --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 2 --aggregator Median)"
echo "mlp 60x20 library linear     $(BLADES_LINEAR_IMPL=lib b --dataset mnist --model mlp --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 2 --aggregator Median)"
echo "resnet 60x20 local steps     $(b --dataset cifar10 --model resnet --clients 60 --batch 64 --local-steps 20 --steps 4 --warmup 1 --aggregator Median)"
done
} 2>&1 | tee $O/general19.log
