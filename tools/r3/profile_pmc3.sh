#!/bin/bash
# PMC passes over tools/r3/kernels_pmc3.py (shipping fp32 kernels at bench scale) -> gpurun_out/r3/pmc/table.txt
# One counter group per run (rocprofv3 does not split passes); no trace domains combined with --pmc.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rm -rf /tmp/pmc3_$name
  timeout -s KILL 170 rocprofv3 --pmc "$@" -d /tmp/pmc3_$name -o p --output-format csv -- python3 $ROOT/tools/r3/kernels_pmc3.py 200 64 > $OUT/$name.run.log 2>&1
  f=$(find /tmp/pmc3_$name -name '*counter_collection.csv' | head -1)
  cp "$f" $OUT/$name.csv
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 $ROOT/tools/r3/pmc_table.py $OUT/sq.csv $OUT/fetch.csv $OUT/write.csv > $OUT/table.txt
rm -f $OUT/*.csv
cat $OUT/table.txt
