"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name.

usage: python tools/pmc_summary.py <counter_collection.csv> [name-substring ...]
"""
import csv
import sys
from collections import defaultdict


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if pats and not any(p in name for p in pats):
                continue
            key = (name[:70], row.get("Grid_Size", ""))
            acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for (name, grid), ctrs in acc.items():
        print(f"{name}  grid={grid}")
        for c, vals in sorted(ctrs.items()):
            print(f"    {c:34s} {sum(vals) / len(vals):16.1f}  (n={len(vals)})")


if __name__ == "__main__":
    main()
