#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean of each counter
per dispatch.  usage: pmc_summary.py <dir-or-csv> [more ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void gs::", "").replace("gs::", "")
    return name[:name.index("(")] if "(" in name else name


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for arg in sys.argv[1:]:
        files = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", acc[k].get("GRBM_GUI_ACTIVE", [0])))):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        vals = {c: sum(v) / len(v) for c, v in acc[k].items()}
        n = len(next(iter(acc[k].values())))
        print("%s  (dispatches %d)" % (k[:110], n))
        print("    " + "  ".join("%s=%.4g" % (c, v) for c, v in sorted(vals.items())))


if __name__ == "__main__":
    main()
