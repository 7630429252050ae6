#!/bin/bash
# Small configurations: one chunk for all 48 training clients (the engine's "auto" rule) vs the bench's fixed 32 (two chunks of 24)
set -o pipefail
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/small_configs_chunk_ab.log
: > $OUT
run() {  # label model dataset localsteps chunk
  echo -n "$1 chunk $5: " >> $OUT
  timeout -k 10 200 python bench.py --model $2 --dataset $3 --clients 60 --malicious-frac 0.2 --local-steps $4 --chunk $5 --steps 10 --warmup 5 2>> gpurun_out/r3/small_configs_chunk_ab.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms')" >> $OUT || exit 1
}
for c in 32 64; do
  run "cnn 60x20" cnn fashionmnist 20 $c && run "mlp 60x20" mlp mnist 20 $c && run "resnet 60x20" resnet cifar10 20 $c && run "cct 60x4" cct cifar10 4 $c \
  && run "mlp 60 FedSGD" mlp mnist 1 $c && run "cnn 60 FedSGD" cnn fashionmnist 1 $c && run "cct 60 FedSGD" cct cifar10 1 $c && run "resnet 60 FedSGD" resnet cifar10 1 $c || exit 1
done
cat $OUT
