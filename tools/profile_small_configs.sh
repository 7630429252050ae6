#!/bin/bash
# Steady-state kernel tables of the reference's 60-client experiment shapes (MLP / CNN / CCT / ResNet-10) on one GPU:
# gpurun_out/small/<model>_kernels.txt + rounds/s lines in gpurun_out/small/bench.log   (run on the GPU box)
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/small
mkdir -p $OUT
: > $OUT/bench.log
export TMPDIR=/tmp
for cfg in "mnist mlp" "fashionmnist cnn" "cifar10 cct" "cifar10 resnet"; do
  set -- $cfg
  timeout -k 10 200 python3 bench.py --dataset "$1" --model "$2" --clients 60 --batch 64 --steps 50 --warmup 5 --aggregator Median 2>>$OUT/bench.err | cut -c1-400 >> $OUT/bench.log
  echo "bench $2 done"
  TRACE=$(mktemp -d /tmp/blades_trace.XXXXXX)
  BLADES_HIP_GRAPH=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 bench.py --dataset "$1" --model "$2" --clients 60 --batch 64 --steps 12 --warmup 3 --aggregator Median > /dev/null 2>>$OUT/prof.err
  python3 tools/steady_stats.py "$TRACE"/*/*_kernel_trace.csv 0.03 25 > $OUT/$2_kernels.txt
  echo "profile $2 done"
done
BLADES_DISABLE_HIP_MHA=1 timeout -k 10 200 python3 bench.py --dataset cifar10 --model cct --clients 60 --batch 64 --steps 50 --warmup 5 --aggregator Median 2>>$OUT/bench.err | cut -c1-400 > $OUT/bench_cct_no_mha.log
