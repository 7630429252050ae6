#!/bin/bash
# round-3 GPU call 21: HBM-side bytes of a whole headline round (FETCH_SIZE / WRITE_SIZE over every dispatch)
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rb_$c
  BLADES_CHUNK_STREAMS=1 timeout -s KILL 280 rocprofv3 --pmc $c -d /tmp/rb_$c -o p --output-format csv -- python3 $ROOT/bench.py --steps 3 --warmup 1 > $O/rb_$c.run.log 2>&1
  f=$(find /tmp/rb_$c -name '*counter_collection.csv' | head -1)
  python3 $ROOT/tools/r3/round_bytes.py "$f" 4 | tee $O/round_bytes_$c.txt
done
