#!/bin/bash
# General path (local steps) on the small models: eager vs HIP-graph replay of a local step, own vs library Linear
set -o pipefail
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/general_path_graph_ab.log
: > $OUT
run() {  # label model dataset steps env...
  label=$1; model=$2; ds=$3; ls=$4; shift 4
  echo -n "$label: " >> $OUT
  env "$@" timeout -k 10 200 python bench.py --model $model --dataset $ds --clients 60 --malicious-frac 0.2 --local-steps $ls --steps 10 --warmup 5 2>> gpurun_out/r3/general_path_graph_ab.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms', d['phase_ms'])" >> $OUT || exit 1
}
run "cnn 60x20 graph auto" cnn fashionmnist 20 X=1 && run "cnn 60x20 graph off " cnn fashionmnist 20 BLADES_HIP_GRAPH=0 && run "cnn 60x20 graph on  " cnn fashionmnist 20 BLADES_HIP_GRAPH=1 \
 && run "cnn 60x20 graph on, lib linear" cnn fashionmnist 20 BLADES_HIP_GRAPH=1 BLADES_LINEAR_IMPL=lib \
 && run "mlp 60x20 graph auto" mlp mnist 20 X=1 && run "mlp 60x20 graph off " mlp mnist 20 BLADES_HIP_GRAPH=0 \
 && run "resnet 60x20 graph auto" resnet cifar10 20 X=1 && run "resnet 60x20 graph on" resnet cifar10 20 BLADES_HIP_GRAPH=1
cat $OUT
