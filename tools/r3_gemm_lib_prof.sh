#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_gemm_lib
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o r -- python3 $R/tools/gemm_vs_library.py > $O/run.txt 2>&1
csv=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - "$csv" > $O/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print(f"calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:400]}")
PY
rm -rf $O/kt
cat $O/kernel_stats.txt
