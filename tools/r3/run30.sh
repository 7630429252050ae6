#!/bin/bash
# PMC passes over the per-client Linear kernels: HBM-side bytes, L2 hit rate, LDS / MFMA activity
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3/pmc_pcl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters...
  n=$1; shift
  rm -rf /tmp/pcl_$n
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d /tmp/pcl_$n -o p --output-format csv -- python3 $ROOT/tools/r3/pc_linear_pmc.py > $O/$n.run.log 2>&1 || return 1
  f=$(find /tmp/pcl_$n -name '*counter_collection.csv' | head -1)
  cp "$f" $O/$n.csv
}
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass l2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum && pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
python3 - $O <<'PY'
import csv, sys, collections
O = sys.argv[1]
acc = collections.OrderedDict()
for n in ("fetch", "write", "l2", "sq"):
    for row in csv.DictReader(open(f"{O}/{n}.csv")):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        if not ("pc_linear" in k or "Cijk" in k):
            continue
        key = (k, row["Grid_Size"])
        acc.setdefault(key, collections.defaultdict(list))[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = lambda d, k: (sum(d[k]) / len(d[k])) if d.get(k) else float("nan")
for (k, g), d in acc.items():
    gui = m(d, "GRBM_GUI_ACTIVE")
    print(f"{k:60s} grid {g:>8s}  fetch {m(d,'FETCH_SIZE')*2/1024:7.1f} MB  write {m(d,'WRITE_SIZE')/1024:6.1f} MB  L2 hit {100*m(d,'TCC_HIT_sum')/m(d,'TCC_REQ_sum'):5.1f} % of {m(d,'TCC_REQ_sum')/1e6:6.2f} M req  "
          f"MFMA busy {100*m(d,'SQ_VALU_MFMA_BUSY_CYCLES')/(gui/8*1024):5.1f} %  LDS confl {100*m(d,'SQ_LDS_BANK_CONFLICT')/max(m(d,'SQ_LDS_IDX_ACTIVE'),1):5.1f} %  wait_LDS {100*m(d,'SQ_WAIT_INST_LDS')/m(d,'SQ_BUSY_CYCLES'):5.1f} %  wait_any {100*m(d,'SQ_WAIT_INST_ANY')/m(d,'SQ_BUSY_CYCLES'):5.1f} %  cycles {gui/8:9.0f}")
PY
