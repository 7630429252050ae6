#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 600 python -u -m pytest tests/test_ops_gpu.py -x -q --timeout 120 --timeout-method thread -k "linear or maxpool or bias_grad" > gpurun_out/r3/pc_linear_tests.log 2>&1 || { tail -30 gpurun_out/r3/pc_linear_tests.log; exit 1; }
tail -2 gpurun_out/r3/pc_linear_tests.log
OUT=gpurun_out/r3/pc_linear_ab.log
: > $OUT
run() {  # label model dataset localsteps env
  echo -n "$1: " >> $OUT
  env $5 timeout -k 10 200 python bench.py --model $2 --dataset $3 --clients 60 --malicious-frac 0.2 --local-steps $4 --steps 10 --warmup 5 2>> gpurun_out/r3/pc_linear_ab.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms')" >> $OUT || exit 1
}
run "cnn 60x20 direct fp32 weights " cnn fashionmnist 20 X=1 && run "cnn 60x20 plane split + igemm " cnn fashionmnist 20 BLADES_PC_LINEAR_MAX_ROWS=0 && run "cnn 60x20 library bmm        " cnn fashionmnist 20 BLADES_LINEAR_IMPL=lib \
&& run "mlp 60x20 direct fp32 weights " mlp mnist 20 X=1 && run "mlp 60x20 plane split + igemm " mlp mnist 20 BLADES_PC_LINEAR_MAX_ROWS=0 && run "mlp 60x20 library bmm        " mlp mnist 20 BLADES_LINEAR_IMPL=lib \
&& run "cct 60x4 igemm (rows > 256)   " cct cifar10 4 X=1 && run "cct 60x4 direct (max rows 1e6)" cct cifar10 4 BLADES_PC_LINEAR_MAX_ROWS=1000000 && run "cct 60x4 library bmm         " cct cifar10 4 BLADES_LINEAR_IMPL=lib \
&& run "cnn 60 FedSGD" cnn fashionmnist 1 X=1
cat $OUT
