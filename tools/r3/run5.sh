#!/bin/bash
# round-3 GPU call 5: conv_igemm loop ablations on the four stride-1 layers (timing only where marked)
set -o pipefail
O=gpurun_out/r3; mkdir -p $O
{
for ab in 0 16 8 24 1 17; do
  echo "== BLADES_IGEMM_ABLATE=$ab  (bit 0: no staging loads, bit 3: no weight staging / barriers inside a chunk, bit 4: 32x32x16 MFMAs; all but 0 give wrong results)"
  BLADES_IGEMM_ABLATE=$ab timeout -k 10 120 python tools/probes/conv_igemm_time.py --npl 3 --n 12800 --layers 0 1 2 3 || exit 1
done
} > $O/igemm_ablate5.log 2>&1 || { tail -20 $O/igemm_ablate5.log; exit 1; }
grep -v amdgpu.ids $O/igemm_ablate5.log
