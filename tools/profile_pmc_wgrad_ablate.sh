#!/bin/bash
# Which LDS instructions of conv_wgrad conflict?  SQ pass with the kernel's diagnostic switches:
#   BLADES_WGRAD_ABLATE=1 no staging after the prologue (no ds_write in the K loop), =2 no fragment reads / MFMA.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ab in 0 1 2; do
  export BLADES_WGRAD_ABLATE=$ab
  rm -rf /tmp/pmc_ab$ab
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pmc_ab$ab -o p --output-format csv -- python3 $ROOT/tools/probes/kernels_pmc.py 20 64 wgrad > $OUT/ablate$ab.run.log 2>&1
  f=$(find /tmp/pmc_ab$ab -name '*counter_collection.csv' | head -1)
  python3 $ROOT/tools/pmc_summary.py "$f" conv_wgrad > $OUT/wgrad_ablate$ab.txt
  echo "ablate $ab done"
done
