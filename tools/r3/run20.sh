#!/bin/bash
# round-3 GPU call 20: kernel table of the CNN general path (60 clients x 20 local steps), own Linear vs library
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
here=$(pwd)
for impl in own lib; do
  TRACE=$(mktemp -d /tmp/blades_trace.XXXXXX)
  cd /tmp
  if [ $impl = lib ]; then export BLADES_LINEAR_IMPL=lib; fi
  BLADES_HIP_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 "$here/bench.py" --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 3 --warmup 1 --aggregator Median > /dev/null 2>$here/$O/prof20_$impl.err || exit 1
  cd "$here"
  python3 tools/steady_stats.py "$TRACE"/*/*_kernel_trace.csv 0.3 22 > $O/cnn_local20_${impl}_kernels.txt
  cut -c1-200 $O/cnn_local20_${impl}_kernels.txt
done
