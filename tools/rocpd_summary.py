#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a CSV like `--stats` prints:
name, calls, total_us, avg_us, pct.  Usage: rocpd_summary.py results.db > profiles/xxx.csv"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for name, calls, total, avg, pct in db.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        short = name if len(name) < 160 else name[:157] + "..."
        w.writerow([short, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])


if __name__ == "__main__":
    main()
