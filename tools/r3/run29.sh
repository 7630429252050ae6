#!/bin/bash
# Kernel sequence of one local step of the CNN general path (graph replay), own and library Linear
set -o pipefail
cd /tmp && export TMPDIR=/tmp
here=$GRAFT_REPO_ROOT
O=$here/gpurun_out/r3
mkdir -p $O
one() {  # tag env
  TRACE=/tmp/trace_$1
  rm -rf $TRACE
  export $2
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 "$here/bench.py" --dataset fashionmnist --model cnn --clients 60 --batch 64 --local-steps 20 --steps 6 --warmup 5 > $O/seq_$1.json 2>$O/seq_$1.err || return 1
  f=$(find $TRACE -name '*kernel_trace.csv' | head -1)
  echo "== $1 ($2)"
  python3 $here/tools/r3/gap_stats.py $f 0.3 6 75
  python3 $here/tools/steady_stats.py $f 0.3 30
}
{ one own X=1 && one lib BLADES_LINEAR_IMPL=lib; } > $O/cnn_step_sequence.log 2>&1
rc=$?
tail -3 $O/cnn_step_sequence.log
exit $rc
