"""Idle-time analysis of a rocprofv3 --kernel-trace CSV: total busy / idle time and which kernels are followed by the
largest gaps.  Usage: python tools/gap_analysis.py <kernel_trace.csv> [skip_first_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ks = ks[int(len(ks) * skip):]                       # drop warm-up
span = ks[-1][1] - ks[0][0]
gaps = defaultdict(lambda: [0, 0])
hist = defaultdict(int)
# a gap is time with NO kernel running on any queue: measured from the latest end seen so far, not from the previous
# kernel's end (a long kernel on one lane overlaps many short ones on the other)
front = ks[0][1]
gap_before = [0] * len(ks)
for i in range(1, len(ks)):
    gap_before[i] = max(0, ks[i][0] - front)
    front = max(front, ks[i][1])
for i, ((s0, e0, n0, _q0), (s1, e1, n1, _q1)) in enumerate(zip(ks, ks[1:])):
    g = gap_before[i + 1]
    key = n0.split("(")[0][:48] + " -> " + n1.split("(")[0][:48]
    gaps[key][0] += g
    gaps[key][1] += 1
    hist[min(int(g / 1000).bit_length(), 12)] += g
idle = sum(gap_before)
print("kernels %d  span %.1f ms  some kernel running %.1f ms (%.1f%%)  idle %.1f ms" % (len(ks), span / 1e6, (span - idle) / 1e6, 100 * (span - idle) / span, idle / 1e6))
print("idle by gap size (us bucket upper bound : ms):", {(1 << k): round(v / 1e6, 2) for k, v in sorted(hist.items())})
for key, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("%8.2f ms  %5d x %7.1f us  %s" % (g / 1e6, c, g / c / 1e3, key))
if len(sys.argv) > 3:
    pat = sys.argv[3]
    shown = 0
    for i in range(3, len(ks) - 3):
        g = gap_before[i + 1]
        if g > 100_000 and pat in ks[i][2] and pat in ks[i + 1][2] and shown < 6:
            shown += 1
            print("---- gap %.0f us" % (g / 1e3))
            for j in range(i - 3, i + 5):
                print("   %s  dur %.1f us  gap_before %.1f us" % (ks[j][2].split("(")[0][:60], (ks[j][1] - ks[j][0]) / 1e3, (ks[j][0] - ks[j - 1][1]) / 1e3))
if len(sys.argv) > 4:
    big = sorted(range(len(ks) - 1), key=lambda i: -gap_before[i + 1])[:int(sys.argv[4])]
    ctx = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    t0 = ks[0][0]
    for i in sorted(big):
        print("gap %7.0f us   %s  ->  %s" % (gap_before[i + 1] / 1e3, ks[i][2].split("(")[0][:50], ks[i + 1][2].split("(")[0][:50]))
        for j in range(max(0, i - ctx), min(len(ks), i + 2 + ctx)) if ctx else ():
            print("      %s t=%10.1f us  dur %8.1f us  queue %s  %s" % ("*" if j == i + 1 else " ", (ks[j][0] - t0) / 1e3, (ks[j][1] - ks[j][0]) / 1e3, ks[j][3], ks[j][2].split("(")[0][:70]))
