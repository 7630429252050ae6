#!/bin/bash
# Effective clock of conv_igemm with and without its staging: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration
set -e
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ab in 0 1; do
  rm -rf /tmp/clk$ab
  BLADES_IGEMM_ABLATE=$ab timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/clk$ab -o p --output-format csv -- python3 $ROOT/tools/probes/conv_igemm_time.py --npl 3 --layers 0 2 --reps 3 > /tmp/clk$ab.log 2>&1
  python3 - /tmp/clk$ab $ab <<'PY'
import csv, glob, sys, collections
d, ab = sys.argv[1], sys.argv[2]
cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'], r['Grid_Size'] if 'Grid_Size' in r else '')
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if 'conv_igemm' not in r['Kernel_Name']: continue
    if r['Dispatch_Id'] in dur:
        ns = dur[r['Dispatch_Id']][0]
        agg[r['Grid_Size']].append((float(r['Counter_Value']) / 8 / ns, ns / 1e6))
for g, v in agg.items():
    v = v[1:]
    print(f"ablate={ab} grid={g}: clock {sum(x[0] for x in v)/len(v):.2f} GHz, {sum(x[1] for x in v)/len(v):.3f} ms (n={len(v)})")
PY
done
