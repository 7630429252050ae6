#!/bin/bash
# Small client chunks on one stream: does a chunk whose activations fit the 256 MB Infinity Cache shorten the round?
set -o pipefail
mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/small_chunk_sweep.log
: > $OUT
run() {  # streams chunk graph
  echo "== streams $1 chunk $2 graph $3" >> $OUT
  BLADES_CHUNK_STREAMS=$1 BLADES_HIP_GRAPH=$3 timeout -k 10 200 python bench.py --steps 6 --warmup 2 --chunk $2 2>> gpurun_out/r3/small_chunk_sweep.err \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms  peak', d['peak_hbm_gb'], 'GB')" >> $OUT || exit 1
}
run 3 32 auto && run 1 32 auto && run 1 16 auto && run 1 8 auto && run 1 8 1 && run 1 4 1 && run 2 8 1 && run 2 16 auto && run 3 32 auto
cat $OUT
