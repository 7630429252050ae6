#!/bin/bash
# PMC rows of the one-tile Gram kernel and the row reductions (end of round 3)
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r3/pmc57
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  rm -rf /tmp/pmc57_$name
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d /tmp/pmc57_$name -o p --output-format csv -- python3 $ROOT/tools/r3/gram_pmc_probe.py > $OUT/$name.run.log 2>&1
  f=$(find /tmp/pmc57_$name -name '*counter_collection.csv' | head -1)
  cp "$f" $OUT/$name.csv
  echo "pass $name done"
}
run sq SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 - $OUT/sq.csv $OUT/fetch.csv $OUT/write.csv > $OUT/table.txt <<'PY'
import csv, sys
from collections import OrderedDict, defaultdict
def load(path):
    acc = OrderedDict()
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        if not any(p in name for p in ("gram", "row_partial")):
            continue
        key = (name.replace("(anonymous namespace)::", "").replace("void ", "")[:52], row.get("Grid_Size", ""))
        acc.setdefault(key, defaultdict(list))[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc
def mean(d, k):
    v = d.get(k); return sum(v) / len(v) if v else float("nan")
sq, fe, wr = (load(p) for p in sys.argv[1:4])
print("MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); LDS conflict % = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;")
print("FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B); us = GRBM_GUI_ACTIVE / 8 / 2.0 GHz (a cycle count, not a timer)")
print(f"{'kernel':52s} {'grid':>9s} {'n':>3s} {'MFMA busy':>9s} {'LDS confl':>9s} {'us@2GHz':>8s} {'fetch MB':>9s} {'write MB':>9s}")
for key, c in sq.items():
    gui = mean(c, "GRBM_GUI_ACTIVE")
    busy = mean(c, "SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) * 100 if gui else float("nan")
    idx = mean(c, "SQ_LDS_IDX_ACTIVE")
    confl = mean(c, "SQ_LDS_BANK_CONFLICT") / idx * 100 if idx else 0.0
    f = mean(fe.get(key, {}), "FETCH_SIZE") * 2 / 1024
    w = mean(wr.get(key, {}), "WRITE_SIZE") / 1024
    print(f"{key[0]:52s} {key[1]:>9s} {len(c['GRBM_GUI_ACTIVE']):3d} {busy:8.1f}% {confl:8.1f}% {gui / 8 / 2000:8.1f} {f:9.1f} {w:9.1f}")
PY
rm -f $OUT/*.csv
cat $OUT/table.txt
