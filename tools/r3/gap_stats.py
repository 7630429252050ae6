"""Idle gaps between consecutive kernels of a rocprofv3 kernel_trace.csv (steady-state tail of the trace).

usage: gap_stats.py <kernel_trace.csv> [frac=0.25] [top=30] [sequence_len=0]
For every dispatch: gap = its start - the latest end of everything dispatched before it (0 if it overlaps).  Sums the
gaps by the kernel that FOLLOWS the gap; with sequence_len > 0 also prints that many consecutive dispatches.
"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
seq = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")
                     .replace("(anonymous namespace)::", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - (t1 - t0) * frac
rows = [r for r in rows if r[0] >= cut]
agg = defaultdict(lambda: [0, 0, 0])
busy = gaps = 0
last_end = rows[0][0]
out = []
for s, e, n in rows:
    g = max(0, s - last_end)
    a = agg[n]
    a[0] += g
    a[1] += 1
    a[2] += e - s
    gaps += g
    busy += e - s
    out.append((g, e - s, n))
    last_end = max(last_end, e)
span = last_end - rows[0][0]
print(f"window {span/1e6:.1f} ms: {len(rows)} dispatches, kernel time {busy/1e6:.1f} ms, idle gaps {gaps/1e6:.1f} ms "
      f"({100*gaps/span:.1f}% of the window), mean gap {gaps/len(rows)/1e3:.1f} us")
print("   gap ms  calls  avg gap us  avg run us  kernel that follows the gap")
for n, (g, c, t) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{g/1e6:9.2f} {c:6d} {g/c/1e3:10.1f} {t/c/1e3:11.1f}  {n[:110]}")
if seq:
    print(f"-- {seq} consecutive dispatches (gap us, run us, kernel)")
    mid = len(out) // 2
    for g, d, n in out[mid: mid + seq]:
        print(f"{g/1e3:8.1f} {d/1e3:8.1f}  {n[:110]}")
