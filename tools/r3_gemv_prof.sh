#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${1:-r3_gemv_prof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python3 $R/tools/gemv_bench.py 4 16 > $O/bench.txt 2>&1
csv=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - "$csv" > $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:7.1f}  max {float(r['MaxNs'])/1e3:7.1f}")
PY
rm -rf $O/kt
grep -v "^W2026\|^E2026" $O/bench.txt | tail -9; cat $O/kernel_stats.txt
