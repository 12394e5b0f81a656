#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_vit
mkdir -p $O
cd $R
timeout -k 10 200 python3 tools/vit_bench.py 4 > $O/alone.txt 2>&1; cat $O/alone.txt | tail -3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python3 $R/tools/vit_bench.py 4 > $O/prof.txt 2>&1
csv=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - "$csv" > $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:45]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} total {float(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
rm -rf $O/kt
cat $O/kernel_stats.txt
