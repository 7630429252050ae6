#!/bin/bash
# rocprofv3 hooks for the FL round: kernel trace of a short bench run, summarised over the
# steady-state tail, and — separately, counters must not share a run with traces on this pool — one PMC pass.
#   tools/profile_round.sh [outdir] [bench args...]
# Examples:
#   tools/profile_round.sh gpurun_out/prof --clients 500 --steps 12 --warmup 1
#   PMC="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" tools/profile_round.sh out --clients 100
set -euo pipefail
OUT=${1:-gpurun_out/prof}; shift || true
mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp}
TRACE=$(mktemp -d /tmp/blades_trace.XXXXXX)
timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d "$TRACE" -- python3 bench.py "$@" | tee "$OUT/bench.json"
python3 tools/steady_stats.py "$TRACE"/*/*_kernel_trace.csv 0.015 40 > "$OUT/steady_kernel_stats.txt"
head -25 "$OUT/steady_kernel_stats.txt"
if [ -n "${PMC:-}" ]; then
  PM=$(mktemp -d /tmp/blades_pmc.XXXXXX)
  # shellcheck disable=SC2086
  timeout -s KILL 600 rocprofv3 --pmc $PMC --output-format csv -d "$PM" -- python3 bench.py "$@" > /dev/null
  python3 tools/pmc_summary.py "$PM"/*/*counter_collection.csv > "$OUT/pmc_summary.txt"
  head -40 "$OUT/pmc_summary.txt"
fi
