"""GPU-timeline gaps between consecutive kernels of one evaluation, from a rocprofv3 --kernel-trace CSV:
   python tools/launch_gaps.py <dir with *_kernel_trace.csv>
Prints, per kernel name, the number of dispatches, the mean duration and the mean idle time on the device BEFORE it
(start - end of the previous dispatch), for the last evaluation-sized run in the trace."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
    rows.sort()
    n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
    rows = rows[-n:]
    gap = defaultdict(list)
    dur = defaultdict(list)
    for (s0, e0, _), (s1, e1, k1) in zip(rows, rows[1:]):
        gap[k1].append(s1 - e0)
        dur[k1].append(e1 - s1)
    tot_gap = sum(sum(v) for v in gap.values())
    span = rows[-1][1] - rows[0][0]
    for k in sorted(gap, key=lambda k: -sum(gap[k])):
        g = sorted(gap[k])
        print("%-62s n=%6d  duration %9.2f us   idle before: mean %7.2f us  median %7.2f us" % (k, len(g), sum(dur[k]) / len(g) / 1e3, sum(g) / len(g) / 1e3, g[len(g) // 2] / 1e3))
    print("span %.2f ms, device idle between dispatches %.2f ms (%.2f %%)" % (span / 1e6, tot_gap / 1e6, 100.0 * tot_gap / span))


main()
