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
ad = [s for s, e, n, q, g in ev if "adamw_" in n]
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

# ---- what the compute stream does while each side stream is busy (the overlap windows of DESIGN.md section 6)
main_q = max(by_q, key=lambda q: sum(e - s for s, e, *_ in by_q[q]))
for q, xs in by_q.items():
    if q == main_q:
        continue
    w0, w1 = min(s for s, *_ in xs), max(e for _, e, *_ in xs)
    inside = [x for x in by_q[main_q] if x[1] > w0 and x[0] < w1]
    busy = sum(min(e, w1) - max(s, w0) for s, e, *_ in inside)
    print(f"\n== window of stream {q}: {(w0 - a) / 1e6:.2f} .. {(w1 - a) / 1e6:.2f} ms of the step ({(w1 - w0) / 1e6:.2f} ms); side busy "
          f"{sum(e - s for s, e, *_ in xs) / 1e6:.2f} ms; compute stream inside it: {len(inside)} launches, busy {busy / 1e6:.2f} ms, idle {(w1 - w0 - busy) / 1e6:.2f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _, g in inside:
        k = (n[n.find("::") + 2:] if n.startswith("void (anonymous") or n.startswith("(anonymous") else n)[:58] + f" g={g}"
        agg[k][0] += 1
        agg[k][1] += min(e, w1) - max(s, w0)
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:72s} {c:5d}  {t / 1e6:8.2f} ms  avg {t / c / 1e3:8.1f} us")
    tail = [x for x in by_q[main_q] if x[0] >= w1]
    stop = next((i for i, x in enumerate(tail) if "sumsq" in x[2] or "adamw" in x[2]), len(tail))
    if stop:
        t_end = tail[stop - 1][1]
        agg2 = collections.defaultdict(lambda: [0, 0])
        for s2, e2, n2, _, g2 in tail[:stop]:
            k2 = (n2[n2.find("::") + 2:] if n2.startswith("void (anonymous") or n2.startswith("(anonymous") else n2)[:58] + f" g={g2}"
            agg2[k2][0] += 1
            agg2[k2][1] += e2 - s2
        print(f"   compute stream AFTER the window, up to the gradient-norm sweep / AdamW: {stop} launches over {(t_end - w1) / 1e6:.2f} ms "
              f"(busy {sum(v[1] for v in agg2.values()) / 1e6:.2f} ms)")
        for k2, (c2, t2) in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:10]:
            print(f"      {k2:72s} {c2:5d}  {t2 / 1e6:8.2f} ms  avg {t2 / c2 / 1e3:8.1f} us")
    after = tail[:3]
    before = [x for x in by_q[main_q] if x[1] <= w0][-2:]
    print("   compute stream just before:", [(n[:40], round((e - s) / 1e3, 1)) for s, e, n, *_ in before])
    print("   compute stream just after: ", [(n[:40], round((s - w1) / 1e3, 1), round((e - s) / 1e3, 1)) for s, e, n, *_ in after])
