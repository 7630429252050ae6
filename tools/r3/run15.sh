#!/bin/bash
# round-3 GPU call 15: serial kernel table + PMC table of the current build
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
bash tools/profile_serial_round.sh $O/serial2 || exit 1
head -24 $O/serial2/steady_kernel_stats.txt
bash tools/r3/profile_pmc3.sh || exit 1
