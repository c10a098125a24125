#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv per step: prof_summary.py <dir> <steps_in_run>"""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = (glob.glob(d + "/*/*kernel_stats.csv") + glob.glob(d + "/*kernel_stats.csv"))[0]
rows = list(csv.DictReader(open(f)))
tot = 0
out = []
for r in rows:
    n = r["Name"]
    if "rocclr" in n: continue
    ms = float(r["TotalDurationNs"]) / 1e6 / steps
    tot += ms
    out.append("%-64s calls/step %6.1f  ms/step %7.3f  avg_us %8.2f" % (n[:64], int(r["Calls"]) / steps, ms, float(r["AverageNs"]) / 1e3))
print("\n".join(out))
print("sum of kernel durations: %.3f ms/step over %d kernel launches/step" % (tot, sum(int(r["Calls"]) for r in rows if "rocclr" not in r["Name"]) / steps))
