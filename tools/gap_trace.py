#!/usr/bin/env python3
"""Idle time between the kernels of one emulated call: reads a rocprofv3 kernel_trace.csv of tools/shape_profile.py and prints, for the
last calls, every kernel's duration and the gap to its predecessor.  usage: tools/gap_trace.py <kernel_trace.csv> [kernels per call]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith(("void oz2::", "oz2::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 10
last = rows[-3 * per:]
t_busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
t_span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
prev = None
for r in last[-per:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} {(e - s) / 1e3:9.1f} us   gap before {gap:7.1f} us")
    prev = e
print(f"last {3 * per} kernels: busy {t_busy / 1e3:.1f} us of {t_span / 1e3:.1f} us span = {100.0 * t_busy / t_span:.1f} %")
