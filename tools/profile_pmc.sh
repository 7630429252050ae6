#!/bin/bash
# PMC passes over tools/probes/kernels_pmc.py -> gpurun_out/pmc_r2/{sq,l2,fetch,write}.txt   (run on the GPU box)
# One counter group per run (rocprofv3 does not split passes); no trace domains combined with --pmc.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc_$name
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d /tmp/pmc_$name -o p --output-format csv -- python3 $ROOT/tools/probes/kernels_pmc.py 20 64 > $OUT/$name.run.log 2>&1
  f=$(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1)
  python3 $ROOT/tools/pmc_summary.py "$f" conv_wgrad conv_igemm gram > $OUT/$name.txt
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
run l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
