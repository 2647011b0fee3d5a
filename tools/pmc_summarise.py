#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: per kernel name (demangled prefix) and counter, the number of
dispatches, the sum and the mean of the counter, and the summed kernel duration (from the
start/end timestamps of the same rows).  Usage: pmc_summarise.py DIR [DIR ...] > summary.md"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("tlpk::", "")
    for cut in ("(", "<"):
        if cut in name:
            name = name[: name.index(cut)]
    return name.replace("void ", "").strip()


def main():
    rows = defaultdict(lambda: [0, 0.0, 0.0])          # (kernel, counter) -> [dispatches, sum, ns]
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = (short(r["Kernel_Name"]), r["Counter_Name"])
                rows[k][0] += 1
                rows[k][1] += float(r["Counter_Value"])
                rows[k][2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print("| kernel | counter | dispatches | sum | mean per dispatch | summed duration (ms) |")
    print("|---|---|---|---|---|---|")
    for (k, c), (n, s, ns) in sorted(rows.items(), key=lambda kv: (-kv[1][2], kv[0])):
        print(f"| {k} | {c} | {n} | {s:.6g} | {s / max(n, 1):.6g} | {ns / 1e6:.3f} |")


if __name__ == "__main__":
    main()
