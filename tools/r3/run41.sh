#!/bin/bash
set -e
mkdir -p gpurun_out
for a in Clippedclustering Signguard Centeredclipping; do
  echo "=== $a" >> gpurun_out/run41.log
  timeout -k 10 200 python -u tools/r3/agg_host_profile.py $a 60 resnet 2>&1 | grep -v "^$" | head -40 >> gpurun_out/run41.log
done
