#!/usr/bin/env python3
"""Per-dispatch counter table from a rocprofv3 --pmc pass (counter_collection.csv):  python tools/pmc_kernels.py <dir>
Groups by (kernel, grid): mean over dispatches of every counter, duration, derived clock (GRBM_GUI_ACTIVE / 8 / ns) and MFMA busy
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)); SQ_* wave counters are quad-cycles summed over waves."""
import collections, csv, glob, sys
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    disp = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:90], r.get("Grid_Size", ""), r["Dispatch_Id"])
        disp[key][r["Counter_Name"]] = disp[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        disp[key]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for (k, gs, _), v in disp.items():
        for c, x in v.items():
            rows[(k, gs)][c].append(x)
for (k, gs), v in sorted(rows.items(), key=lambda kv: -sum(kv[1]["_ns"])):
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    n = len(v["_ns"])
    mean = {c: sum(x[n // 3:]) / max(1, len(x[n // 3:])) for c, x in v.items()}      # skip the first third (warm-up launches)
    ns = mean["_ns"]
    clk = mean.get("GRBM_GUI_ACTIVE", 0) / 8.0 / ns if ns else 0
    print(f"== {k}  grid {gs}  dispatches {n}  {ns / 1e3:.1f} us  clock {clk:.3f} GHz")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and clk:
        print(f"   MFMA busy {100 * mean['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * clk * ns):.1f} %")
    wc = mean.get("SQ_WAVE_CYCLES", 0)
    for c in sorted(mean):
        if c.startswith("_") or c in ("GRBM_GUI_ACTIVE",):
            continue
        extra = f"  ({100 * mean[c] / wc:.1f} % of wave cycles)" if wc and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES") else ""
        print(f"   {c:28s} {mean[c]:16.0f}{extra}")
