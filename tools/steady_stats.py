"""Steady-state kernel summary from a rocprofv3 kernel_trace.csv.

MIOpen's find phase (first round) benchmarks dozens of candidate kernels and swamps
`--stats`; this keeps only the dispatches of the last `frac` of the traced time span.
usage: steady_stats.py <kernel_trace.csv> [frac=0.25] [top=40]
"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
cut = t1 - (t1 - t0) * frac
agg = defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in rows:
    if s >= cut:
        agg[n][0] += e - s
        agg[n][1] += 1
        busy += e - s
span = t1 - cut
print(f"window {span/1e6:.1f} ms, kernel-busy {busy/1e6:.1f} ms ({100*busy/span:.1f}% of wall)")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{t/1e6:9.2f} ms {100*t/busy:5.1f}% calls {c:6d} avg {t/c/1e3:9.1f} us  {n.replace('void ', '')[:120]}")
