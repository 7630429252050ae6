#!/bin/bash
# round-3 GPU call 33: validation of the current build: full GPU suite, RF-everywhere subset, smoke, 20-round headline, small configs, general path
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > $O/pytest33.log 2>&1
rc=$?; tail -3 $O/pytest33.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest33.log | head -40; exit 1; fi
BLADES_IGEMM_RF=1 timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "conv_igemm or linear_forward or conv_epilogue or conv_dgrad or stem_as" --timeout 120 --timeout-method thread > $O/pytest33b.log 2>&1
rc=$?; tail -2 $O/pytest33b.log
if [ $rc -ne 0 ]; then grep -n "Error\|FAILED\|assert" $O/pytest33b.log | head -30; exit 1; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 bench.py --steps 20 --warmup 2 > $O/bench33_20rounds.log 2>$O/bench33.err || { tail $O/bench33.err; exit 1; }
cut -c1-260 $O/bench33_20rounds.log
b() { timeout -k 10 200 python3 bench.py "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])'; }
small="--clients 60 --batch 64 --steps 50 --warmup 5 --aggregator Median"
{
for cfg in "mnist mlp" "fashionmnist cnn" "cifar10 cct" "cifar10 resnet"; do
  set -- $cfg
  echo "$2 60 clients FedSGD        $(b --dataset $1 --model $2 $small)"
done
echo "cnn 60x20 local steps        $(b --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 10 --warmup 5 --aggregator Median)"
echo "mlp 60x20 local steps        $(b --dataset mnist --model mlp --clients 60 --batch 64 --local-steps 20 --steps 10 --warmup 5 --aggregator Median)"
echo "cct 60x4 local steps         $(b --dataset cifar10 --model cct --clients 60 --batch 64 --local-steps 4 --steps 10 --warmup 5 --aggregator Median)"
echo "resnet 60x20 local steps     $(b --dataset cifar10 --model resnet --clients 60 --batch 64 --local-steps 20 --steps 5 --warmup 4 --aggregator Median)"
echo "resnet18 bf16x3 headline     $(b --amp bf16x3 --steps 6 --warmup 2)"
echo "resnet18 bf16 headline       $(b --amp bf16 --steps 6 --warmup 2)"
} 2>&1 | tee $O/configs33.log
