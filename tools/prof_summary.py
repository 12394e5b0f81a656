#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv:  python tools/prof_summary.py <csv> <steps-in-trace> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / steps:.1f} ms/step")
for r in rows[:top]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6 / steps:8.2f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
