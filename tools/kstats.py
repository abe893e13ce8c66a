#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 *kernel_stats.csv (name, calls, average ms, total ms).  usage: tools/kstats.py <csv> [rows=14]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]
for r in rows:
    print(f"{r['Name'][:72]:72s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']) / 1e6:8.3f} ms  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
