#!/bin/bash
# PMC passes over tools/probes/kernels_pmc_strided.py -> gpurun_out/pmc_strided/{sq,fetch}.txt   (run on the GPU box)
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_strided
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmcs_$name
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d /tmp/pmcs_$name -o p --output-format csv -- python3 $ROOT/tools/probes/kernels_pmc_strided.py 20 64 > $OUT/$name.run.log 2>&1
  f=$(find /tmp/pmcs_$name -name '*counter_collection.csv' | head -1)
  python3 $ROOT/tools/pmc_summary.py "$f" conv_wgrad conv_igemm > $OUT/$name.txt
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
