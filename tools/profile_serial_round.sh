#!/bin/bash
# Kernel trace of the fp32 headline round with ONE chunk stream (un-overlapped kernel durations = their true cost),
# summarised over the last half of the run:  tools/profile_serial_round.sh [outdir]
set -euo pipefail
OUT=${1:-gpurun_out/prof_serial}
mkdir -p "$OUT"
export TMPDIR=/tmp
export BLADES_CHUNK_STREAMS=1
TRACE=$(mktemp -d /tmp/blades_trace.XXXXXX)
here=$(pwd)
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 "$here/bench.py" --steps 4 --warmup 2 > "$here/$OUT/bench.json"
cd "$here"
python3 tools/steady_stats.py "$TRACE"/*/*_kernel_trace.csv 0.5 40 > "$OUT/steady_kernel_stats.txt"
