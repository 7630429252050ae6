"""Sum a PMC counter over every dispatch of a bench run: python tools/r3/round_bytes.py <counter_collection.csv> <rounds>"""
import csv, sys
from collections import defaultdict
tot = defaultdict(float); per = defaultdict(float)
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        v = float(row["Counter_Value"]); tot[row["Counter_Name"]] += v
        per[(row["Counter_Name"], row["Kernel_Name"].replace("(anonymous namespace)::", "")[:60])] += v
rounds = float(sys.argv[2])
for c, v in tot.items():
    scale = 2.0 if c == "FETCH_SIZE" else 1.0   # gfx950 tallies 128-B read requests at 64 B (guide); KB units
    print(f"{c}: {v * scale / 1e6 / rounds:.1f} GB per round (raw sum {v:.0f} KB over {rounds:.0f} rounds incl. warm-up)")
    top = sorted(((vv, k[1]) for k, vv in per.items() if k[0] == c), reverse=True)[:8]
    for vv, name in top:
        print(f"    {vv * scale / 1e6 / rounds:8.1f} GB  {name}")
