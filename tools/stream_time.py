#!/usr/bin/env python3
"""Per-stream composition of ONE training step from a rocprofv3 --kernel-trace of bench.py:
   python tools/stream_time.py <kernel_trace.csv>
The step window = between the first AdamW launches of the last two optimiser bursts.  For every HIP stream (queue) in it:
busy time, and per kernel (name + grid) launches / total / average.  The stream with the most busy time is the compute
stream: its total is the step's critical path (the other streams run beside it)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
qk = "Stream_Id" if "Stream_Id" in rows[0] else "Queue_Id"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[qk], r.get("Grid_Size_X", r.get("Grid_Size", "?"))) for r in rows))
ad = [s for s, e, n, q, g in ev if "adamw_kernel" in n]
bounds = [ad[0]] + [ad[i] for i in range(1, len(ad)) if ad[i] - ad[i - 1] > 100e6]
a, b = bounds[-2], bounds[-1]
win = [x for x in ev if a <= x[0] < b]
print(f"step window {(b - a) / 1e6:.2f} ms, {len(win)} launches, streams keyed by {qk}")
by_q = collections.defaultdict(list)
for x in win:
    by_q[x[3]].append(x)
for q, xs in sorted(by_q.items(), key=lambda kv: -sum(e - s for s, e, *_ in kv[1])):
    busy = sum(e - s for s, e, *_ in xs)
    print(f"\n== stream {q}: {len(xs)} launches, busy {busy / 1e6:.2f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _, g in xs:
        k = (n[n.find("::") + 2:] if n.startswith("void (anonymous") or n.startswith("(anonymous") else n)[:58] + f" g={g}"
        agg[k][0] += 1
        agg[k][1] += e - s
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {k:72s} {c:5d}  {t / 1e6:8.2f} ms  avg {t / c / 1e3:8.1f} us")
