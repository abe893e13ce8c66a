#!/usr/bin/env python3
"""Kernel timeline of the headline step: durations and the idle time in front of every kernel of two consecutive timed steps, and busy vs span per step.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace -d /tmp/gp -o t --output-format csv -- python $REPO/bench.py --no-cpu --steps 6 --warmup 2; python tools/step_gaps.py /tmp/gp [kernels per step = 9]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 9
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("void oz2::", "oz2::"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
prev = None
for r in rows[per * 3:per * 5]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:60]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{name:60s} {(e - s) / 1e3:8.1f} us   idle before {gap:6.1f} us")
    prev = e
a = rows[per * 2:per * 8]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in a) / 6e3
span = (int(rows[per * 8]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"])) / 6e3
print(f"six timed steps: busy {busy:.1f} us / step, span {span:.1f} us / step, idle {span - busy:.1f} us = {100 * (span - busy) / span:.2f} %")
