"""Print the top-N rows of a rocprofv3 kernel_stats.csv (short names)."""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = [r for r in csv.DictReader(open(path)) if not r["Name"].startswith("naive_conv")]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {len(rows)} kernels")
for r in rows[:n]:
    name = r["Name"].replace("void ", "")[:110]
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us  {name}')
