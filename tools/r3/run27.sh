#!/bin/bash
# Where are the idle gaps of a graph-replayed small round?  (CNN 60 x 20 local steps, one chunk; MLP 60 x 20; CNN FedSGD)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
here=$GRAFT_REPO_ROOT
O=$here/gpurun_out/r3
mkdir -p $O
one() {  # tag model dataset localsteps graph
  TRACE=/tmp/trace_$1
  rm -rf $TRACE
  BLADES_HIP_GRAPH=$5 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 "$here/bench.py" --dataset $3 --model $2 --clients 60 --batch 64 --local-steps $4 --steps 6 --warmup 5 > $O/gaps_$1.json 2>$O/gaps_$1.err || return 1
  f=$(find $TRACE -name '*kernel_trace.csv' | head -1)
  echo "== $1 (graph $5): $(python3 -c "import json,sys; print(json.loads(open('$O/gaps_$1.json').readlines()[-1])['ms_per_step'])") ms per round under the profiler"
  python3 $here/tools/r3/gap_stats.py $f 0.3 25 70
}
{ one cnn20_graph cnn fashionmnist 20 1 && one cnn20_eager cnn fashionmnist 20 0 && one mlp20_graph mlp mnist 20 1 && one cnn1_graph cnn fashionmnist 1 1; } > $O/graph_gaps.log 2>&1
rc=$?
tail -5 $O/graph_gaps.log
exit $rc
