#!/bin/bash
# round-3 GPU call 22: L2 hit rates of the shipping conv kernels at bench scale
set -o pipefail
ROOT=$(pwd); O=$ROOT/gpurun_out/r3/pmc; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc3_l2
timeout -s KILL 170 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d /tmp/pmc3_l2 -o p --output-format csv -- python3 $ROOT/tools/r3/kernels_pmc3.py 200 64 > $O/l2.run.log 2>&1
f=$(find /tmp/pmc3_l2 -name '*counter_collection.csv' | head -1)
python3 $ROOT/tools/pmc_summary.py "$f" conv_igemm conv_wgrad > $O/l2.txt
python3 - "$O/l2.txt" <<'PY'
import sys, re
name=None; vals={}
for l in open(sys.argv[1]):
    if not l.startswith(' '):
        if name and vals: print(f"{name[:60]:60s} hit {100*vals.get('TCC_HIT_sum',0)/max(vals.get('TCC_REQ_sum',1),1):5.1f} %  req {vals.get('TCC_REQ_sum',0)/1e6:8.1f} M  miss {vals.get('TCC_MISS_sum',0)/1e6:8.1f} M")
        name=l.strip(); vals={}
    else:
        p=l.split(); vals[p[0]]=float(p[1])
if name and vals: print(f"{name[:60]:60s} hit {100*vals.get('TCC_HIT_sum',0)/max(vals.get('TCC_REQ_sum',1),1):5.1f} %  req {vals.get('TCC_REQ_sum',0)/1e6:8.1f} M  miss {vals.get('TCC_MISS_sum',0)/1e6:8.1f} M")
PY
