#!/usr/bin/env python3
"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace run of bench.py:
   python tools/gap_analysis.py <kernel_trace.csv> [min_gap_us]
Merges the busy intervals of all streams, reports idle time between the first and the last AdamW launch of consecutive
steps (one optimiser launch burst per step marks the step boundary) and the largest gaps with their neighbours."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# step boundaries = first adamw kernel of each burst (bursts are > 100 ms apart)
ad = [s for s, e, n in ev if "adamw_kernel" in n]
bounds = [ad[0]] + [ad[i] for i in range(1, len(ad)) if ad[i] - ad[i - 1] > 100e6]
print("optimizer bursts:", len(bounds))
for a, b in zip(bounds[:-1], bounds[1:]):
    win = [(s, e, n) for s, e, n in ev if s >= a and s < b]
    busy_end = win[0][1]
    idle = 0
    gaps = []
    prev = win[0][2]
    for s, e, n in win[1:]:
        if s > busy_end:
            idle += s - busy_end
            if (s - busy_end) / 1e3 >= thr:
                gaps.append(((s - busy_end) / 1e3, prev[:50], n[:50]))
        if e > busy_end:
            busy_end = e
            prev = n
    print(f"step window {(b - a) / 1e6:.1f} ms: idle {idle / 1e6:.2f} ms in {len(win)} launches; gaps >= {thr} us: {len(gaps)}")
    for g in sorted(gaps, reverse=True)[:12]:
        print(f"   {g[0]:8.1f} us  after {g[1]}  before {g[2]}")
