#!/bin/bash
# bf16 conv_igemm: what bounds it?  timing-only ablations (wrong results) of the per-tap kernel
set -e
mkdir -p gpurun_out
for a in 0 1 2 4 8 9 13; do
  echo "ablate=$a" | tee -a gpurun_out/run44.log
  BLADES_IGEMM_CR=0 BLADES_IGEMM_ABLATE=$a timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 --amp bf16 2>&1 | tail -1 | cut -c1-140 | tee -a gpurun_out/run44.log
done
