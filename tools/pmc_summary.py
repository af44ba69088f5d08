#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch.
Usage: pmc_summary.py <dir with *counter_collection.csv files> > summary.csv"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row.get("Kernel_Name", "?").split("(")[0][:60]
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    w = csv.writer(sys.stdout)
    counters = sorted({c for k in acc for c in acc[k]})
    w.writerow(["kernel", "dispatches"] + counters)
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0]))):
        n = max(len(v) for v in acc[k].values())
        w.writerow([k, n] + ["%.4g" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "" for c in counters])


if __name__ == "__main__":
    main()
