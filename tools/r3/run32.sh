#!/bin/bash
# End-to-end: all nine aggregators under ALIE with the round-3 build (synthetic CIFAR-10 stand-in), through the CLI
set -o pipefail
mkdir -p gpurun_out/r3/e2e
export BLADES_SYNTHETIC_NOISE=3
( while true; do sleep 50; echo "[alive $(date +%T)]"; done ) &
HB=$!
timeout -k 10 1000 python -m blades_amd.train file examples/alie_resnet10_synthetic_all_aggregators.yaml --synthetic-data --local-mode --local-dir gpurun_out/r3/e2e > gpurun_out/r3/e2e_all_aggregators.log 2>&1
rc=$?
kill $HB
tail -25 gpurun_out/r3/e2e_all_aggregators.log | cut -c1-250
exit $rc
