#!/bin/bash
# round-3 GPU call 3: full GPU suite with the new kernels (Linear GEMM, elementwise, per-client LN, im2col wgrad), small-config A/B
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread > $O/pytest3.log 2>&1
rc=$?; tail -5 $O/pytest3.log
if [ $rc -ne 0 ]; then grep -n "FAILED" $O/pytest3.log | head -40; fi
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
small="--clients 60 --batch 64 --steps 50 --warmup 5 --aggregator Median"
OLD="BLADES_LINEAR_IMPL=lib BLADES_FUSED_CE=0"
{
for i in 1 2; do
for cfg in "mnist mlp" "fashionmnist cnn" "cifar10 cct" "cifar10 resnet"; do
  set -- $cfg
  echo "$2 new                  $(b --dataset $1 --model $2 $small)"
  echo "$2 library linear + vmap loss  $(env $OLD python3 -c 'print(1)' >/dev/null; BLADES_LINEAR_IMPL=lib BLADES_FUSED_CE=0 b --dataset $1 --model $2 $small)"
done
done
echo "general path (60 x 20 local steps)"
echo "cnn 60x20 new            $(b --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 2 --aggregator Median)"
echo "cnn 60x20 library linear $(BLADES_LINEAR_IMPL=lib b --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 2 --aggregator Median)"
echo "cct 60x4 new             $(b --dataset cifar10 --model cct --clients 60 --batch 64 --local-steps 4 --steps 6 --warmup 2 --aggregator Median)"
echo "resnet 60x20 new         $(b --dataset cifar10 --model resnet --clients 60 --batch 64 --local-steps 20 --steps 4 --warmup 1 --aggregator Median)"
} 2>&1 | tee $O/small_ab3.log
