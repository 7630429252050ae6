#!/bin/bash
# Steady-state kernel tables of the small FedSGD configurations (one chunk, graph replay): what is left besides MFMA work?
set -o pipefail
cd /tmp && export TMPDIR=/tmp
here=$GRAFT_REPO_ROOT
O=$here/gpurun_out/r3
mkdir -p $O
one() {  # tag model dataset localsteps
  TRACE=/tmp/trace_$1
  rm -rf $TRACE
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 "$here/bench.py" --dataset $3 --model $2 --clients 60 --batch 64 --local-steps $4 --steps 8 --warmup 5 > $O/tab_$1.json 2>$O/tab_$1.err || return 1
  f=$(find $TRACE -name '*kernel_trace.csv' | head -1)
  echo "== $1: $(python3 -c "import json; print(json.loads(open('$O/tab_$1.json').readlines()[-1])['ms_per_step'])") ms per round under the profiler"
  python3 $here/tools/r3/gap_stats.py $f 0.3 4 0 | head -2
  python3 $here/tools/steady_stats.py $f 0.3 28
}
{ one cct_fedsgd cct cifar10 1 && one resnet_fedsgd resnet cifar10 1 && one cnn_fedsgd cnn fashionmnist 1 && one mlp_fedsgd mlp mnist 1 && one cct_4 cct cifar10 4; } > $O/small_tables.log 2>&1
rc=$?
tail -3 $O/small_tables.log
exit $rc
