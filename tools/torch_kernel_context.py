#!/usr/bin/env python3
"""Where do torch's own kernels (fills, copies, elementwise) sit inside a training step?
   python tools/torch_kernel_context.py <kernel_trace.csv>
For the LAST complete step window of a rocprofv3 --kernel-trace of bench.py: every kernel that is not a libmmhip kernel,
grouped by (previous libmmhip kernel, name, workgroups) with count and total time -- the memory plumbing that DESIGN.md
says is not on the path."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
              int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1))
             for r in rows), key=lambda e: e[0])
ad = [s for s, e, n, g, w in ev if "adamw_kernel" in n]
bounds = [ad[0]] + [ad[i] for i in range(1, len(ad)) if ad[i] - ad[i - 1] > 100e6]
a, b = bounds[-2], bounds[-1]
win = [x for x in ev if a <= x[0] < b]
ours = lambda n: "anonymous namespace" in n or "_GLOBAL__N_" in n  # noqa: E731
agg = collections.OrderedDict()
prev = "-"
for s, e, n, g, w in win:
    if ours(n) and "at::native" not in n:
        prev = n.split("(")[0][-40:]
        continue
    key = (prev, n[:70], g // max(w, 1))
    c = agg.setdefault(key, [0, 0.0])
    c[0] += 1
    c[1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"step window {(b - a) / 1e6:.1f} ms; non-libmmhip kernels: {sum(v[0] for v in agg.values())} launches, {tot / 1e3:.2f} ms")
for (p, n, wg), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{cnt:5d} x {us / cnt:8.1f} us = {us / 1e3:6.2f} ms  wgs {wg:8d}  {n:70s} after {p}")
