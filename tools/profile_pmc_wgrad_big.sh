#!/bin/bash
# MFMA busy cycles of conv_wgrad at bench scale (200 clients x 64): cycle-based utilisation, independent of the clock
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_big
timeout -s KILL 170 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_big -o p --output-format csv -- python3 $ROOT/tools/probes/kernels_pmc.py 200 64 wgrad > $OUT/big.run.log 2>&1
f=$(find /tmp/pmc_big -name '*counter_collection.csv' | head -1)
python3 $ROOT/tools/pmc_summary.py "$f" conv_wgrad > $OUT/wgrad_big.txt
cat $OUT/wgrad_big.txt
