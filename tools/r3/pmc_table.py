"""Derived table from the three PMC passes of tools/r3/profile_pmc3.sh.
usage: python tools/r3/pmc_table.py sq.csv fetch.csv write.csv"""
import csv
import sys
from collections import OrderedDict, defaultdict


def load(path):
    acc = OrderedDict()
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if not any(p in name for p in ("conv_wgrad", "conv_igemm", "gram")):
                continue
            key = (name.replace("(anonymous namespace)::", "").replace("void ", "")[:44], row.get("Grid_Size", ""))
            acc.setdefault(key, defaultdict(list))[row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def mean(d, k):
    v = d.get(k)
    return sum(v) / len(v) if v else float("nan")


sq, fe, wr = load(sys.argv[1]), load(sys.argv[2]), load(sys.argv[3])
print("MFMA busy % = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); LDS conflict % = SQ_LDS_BANK_CONFLICT /")
print("SQ_LDS_IDX_ACTIVE; wait_LDS = SQ_WAIT_INST_LDS / SQ_BUSY_CYCLES; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B).")
print(f"{'kernel':44s} {'grid':>9s} {'launches':>8s} {'MFMA busy':>9s} {'LDS confl':>9s} {'wait_LDS':>9s} {'LDS insts M':>11s} {'fetch MB':>9s} {'write MB':>9s}")
for key, c in sq.items():
    gui = mean(c, "GRBM_GUI_ACTIVE")
    busy = mean(c, "SQ_VALU_MFMA_BUSY_CYCLES") / (gui / 8 * 1024) * 100 if gui else float("nan")
    act = mean(c, "SQ_LDS_IDX_ACTIVE")
    confl = mean(c, "SQ_LDS_BANK_CONFLICT") / act * 100 if act else 0.0
    wl = mean(c, "SQ_WAIT_INST_LDS") / mean(c, "SQ_BUSY_CYCLES") * 100
    f = mean(fe.get(key, {}), "FETCH_SIZE") * 2 / 1024
    w = mean(wr.get(key, {}), "WRITE_SIZE") / 1024
    print(f"{key[0]:44s} {key[1]:>9s} {len(c['GRBM_GUI_ACTIVE']):8d} {busy:8.1f}% {confl:8.1f}% {wl:8.1f}% "
          f"{mean(c, 'SQ_INSTS_LDS') / 1e6:11.2f} {f:9.1f} {w:9.1f}")
