#!/bin/bash
# fp32 headline: timing-only ablations of conv_igemm (wrong results): where is the floor without its K loop?
set -e
mkdir -p gpurun_out
rm -f gpurun_out/run45.log
for a in 0 4 13; do
  echo "ablate=$a" | tee -a gpurun_out/run45.log
  BLADES_IGEMM_ABLATE=$a timeout -k 10 300 python -u bench.py --steps 3 --warmup 1 2>&1 | tail -1 | cut -c1-140 | tee -a gpurun_out/run45.log
done
