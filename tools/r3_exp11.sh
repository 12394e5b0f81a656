#!/bin/bash
# round-3 experiment 11: full suite + driver-style bench, then a finer-grained A/B of MM_LOSS_ROWS (the box of exp10 drifted)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r3_exp11
mkdir -p $O
cd $R
bash tools/r3_suite.sh r3_exp11 || exit 1
timeout -k 10 900 python3 tools/step_ab.py --rounds 6 --steps 5 --warmup 2 "MM_LOSS_ROWS=0" "MM_LOSS_ROWS=1" > $O/step_ab.txt 2>&1
tail -3 $O/step_ab.txt
