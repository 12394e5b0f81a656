#!/bin/bash
# Run on the GPU box (through gpurun): PMC passes over tools/attn_bench.py for the D=128 attention kernels.
#   tools/attn_pmc.sh TAG [attn_bench args]
set -e -o pipefail
tag=${1:-attn}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $O/p$i -o r -- python3 $R/tools/attn_bench.py "$@" > $O/p$i.log 2>&1 || true
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); ns = collections.defaultdict(float)
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "attn" not in k: continue
        k = k[k.index("attn"):].split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT"):
            seen.add(key); n[(k, r["Counter_Name"])] += 1; ns[(k, r["Counter_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
with open("$O/summary.txt", "w") as out:
    for k, v in agg.items():
        d = n[(k, "SQ_WAVE_CYCLES")] or 1
        out.write(f"{k}: launches {d}, avg {ns[(k,'SQ_WAVE_CYCLES')]/d/1e3:.1f} us\n")
        for c, x in sorted(v.items()):
            out.write(f"    {c:32s} {x/d:16.0f} per launch\n")
print(open("$O/summary.txt").read())
PY
find $O -name "*.csv" -delete
